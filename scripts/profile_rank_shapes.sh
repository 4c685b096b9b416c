#!/bin/bash
# Rank-stage profiles per DNN3 hidden shape (runs on the GPU box via gpurun): kernel-trace stats, then PMC passes (one
# counter group per run, --pmc only, program directly after `--`) of scripts/dev/rank_shapes.py; writes
# gpurun_out/prof_rank_shapes/{summary.txt,r3_rank_shapes_pmc.json} — copy both into profiles/.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_rank_shapes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/scripts/dev/rank_shapes.py" 100000000 > "$OUT/trace.log" 2>&1
for grp in "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/scripts/dev/rank_shapes.py" 100000000 > "$OUT/pmc_$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import sys, os, csv, glob, collections, json, re
out = sys.argv[1]
SIMDS = 1024
lines = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats: rocprofv3 --kernel-trace --stats -- python3 scripts/dev/rank_shapes.py 100000000")
    for r in csv.DictReader(open(p)):
        lines.append("%-120s calls %6s avg_us %10.1f total_ms %10.2f pct %s" % (r["Name"][:120], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r.get("Percentage", "")))
lines.append(open(os.path.join(out, "trace.log")).read()[-2500:])
agg = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            kn = row.get("Kernel_Name", "?")
            if "dnn3_rs_kernel" in kn or "dnn3_ws_kernel" in kn or "dnn3_ls_kernel" in kn or "mlp_kernel" in kn:
                k = (kn[:70], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
lines.append("== PMC (average per dispatch)")
per = collections.defaultdict(dict)
for (kn, cn), (v, n) in sorted(agg.items()):
    lines.append("%-72s %-28s avg=%.6g  n=%d" % (kn, cn, v / max(n, 1), n))
    per[kn][cn] = v / max(n, 1)
res = {"_how": "rocprofv3 --pmc <group> -- python3 scripts/dev/rank_shapes.py 100000000 (scripts/profile_rank_shapes.sh; one counter group "
               "per run); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of the MLP kernel's launches; "
               "fetch_bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half the bytes of 16 B/lane reads, MI355X_MICROARCH.md); "
               "1.28 M items = 655 MB of table rows"}
def shape_of(kn):
    m = re.search(r"dnn3_[rl]s_kernel<(\d+), (\d+)", kn)
    if m: return "%s-%s" % (m.group(1), m.group(2))
    if "dnn3_ws_kernel" in kn: return "512-256"
    m = re.search(r"mlp_kernel<1, (\d+), (\d+)", kn)
    if m: return "%s-%s%s" % (m.group(1), m.group(2), "" if (m.group(1), m.group(2)) == ("1024", "512") else "-streaming")
    return None
for kn, c in per.items():
    sh = shape_of(kn)
    if not sh: continue
    e = {"kernel": kn}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if "FETCH_SIZE" in c:
        e["fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        e["fetch_over_rows"] = round(e["fetch_bytes"] / (1280000 * 512), 3)
    if c.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    if c.get("SQ_WAVE_CYCLES"):
        e["wave_wait_frac"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
    res[sh] = e
json.dump(res, open(os.path.join(out, "r3_rank_shapes_pmc.json"), "w"), indent=1)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
print(json.dumps(res, indent=1)[:4000])
PY
