#!/bin/bash
# Round-4 profile of the headline step in its steady state (thresholds from the table's model, one screened launch per
# pass).  Runs on the GPU box via gpurun; the program directly after `--`; PMC groups in runs of their own.
# Usage: scripts/profile_r4_headline.sh <tag> [bench args...]
# The summary lists the LAST `TAIL` dispatches of each scan-stage kernel one by one (the first batches of the run are the
# calibration on the pilot plan and would blur an average over all dispatches).
set -u
TAG=${1:-r3h}; shift || true
TAIL=${TAIL:-6}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --latency-reqs 0 --no-extras --no-rank-shapes --no-f32-leg --contexts 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_$name.log" 2>&1
done
python3 - "$OUT" "$TAIL" <<'PY'
import sys, os, csv, glob, collections, json
out, tail = sys.argv[1], int(sys.argv[2])
KERNELS = ("screen_kernel", "screen_decode_kernel", "rescore_kernel", "scan_kernel", "select_kernel", "final_kernel", "dnn3_ws_kernel", "sort_kernel", "pred_")
res = {"_how": "rocprofv3 --kernel-trace --stats / --pmc <group> -- python3 bench.py --steps 6 --warmup 2 --no-extras --contexts 1 "
               "(scripts/profile_r4_headline.sh); per kernel: the last %d dispatches of the run = steady state "
               "(thresholds predicted by the table's model, one screened launch per 256-query pass); "
               "fetch_bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half the bytes of 16 B/lane reads, MI355X_MICROARCH.md), "
               "write_bytes = WRITE_SIZE (KB) x 1024" % tail}
lines = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats, whole run incl. calibration batches (%s)" % os.path.relpath(p, out))
    lines.append(open(p).read())
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        kn = r["Kernel_Name"]
        for k in KERNELS:
            if k in kn:
                by[kn[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines.append("== last %d dispatches per kernel (us), kernel trace" % tail)
    for kn, v in sorted(by.items()):
        t = v[-tail:]
        lines.append("%-82s n=%4d  last: %s  mean %.1f" % (kn, len(v), " ".join("%.1f" % x for x in t), sum(t) / len(t)))
        res.setdefault("kernel_us", {})[kn] = {"dispatches": len(v), "last_mean_us": sum(t) / len(t)}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(p)))
        by = collections.defaultdict(list)
        for r in rows:
            kn = r.get("Kernel_Name", "?")
            if any(k in kn for k in ("screen_kernel", "screen_decode_kernel", "rescore_kernel", "dnn3_ws_kernel")):
                by[(kn[:80], r.get("Counter_Name", "?"))].append((int(r.get("Dispatch_Id", 0)), float(r.get("Counter_Value", 0) or 0)))
        lines.append("== PMC %s: mean over the last %d dispatches" % (os.path.basename(d), tail))
        for (kn, cn), v in sorted(by.items()):
            v.sort()
            t = [x[1] for x in v[-tail:]]
            m = sum(t) / len(t)
            lines.append("%-82s %-28s %.6g" % (kn, cn, m))
            res.setdefault("pmc", {}).setdefault(kn, {})[cn] = m
for kn, c in res.get("pmc", {}).items():
    if "FETCH_SIZE" in c:
        c["fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        c["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("\n".join(lines)[-7000:])
PY
