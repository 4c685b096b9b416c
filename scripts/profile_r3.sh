#!/bin/bash
# Round-3 profiles (runs on the GPU box via gpurun): kernel-trace stats of the coalescer legs, the group step with DPP and
# cfg 4 over item records; PMC passes (one group per run, --pmc only) for the cfg-4 kernel.  Program directly after `--`.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_r3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/callers" -o t -- python3 "$REPO/scripts/dev/callers.py" > "$OUT/callers.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5" -o t -- python3 "$REPO/scripts/dev/cfg5.py" > "$OUT/cfg5.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg4" -o t -- python3 "$REPO/scripts/dev/cfg4c.py" > "$OUT/cfg4.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/cfg4_pmc_$name" -o pmc -- python3 "$REPO/scripts/dev/cfg4c.py" 20000000 > "$OUT/cfg4_pmc_$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import sys, os, csv, glob, collections
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for tag in ("callers", "cfg5", "cfg4"):
        for p in glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True):
            f.write("== kernel stats: %s\n" % tag)
            for r in csv.DictReader(open(p)):
                f.write("%-110s calls %6s avg_us %10.1f total_ms %10.2f pct %s\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r.get("Percentage", "")))
        f.write(open(os.path.join(out, tag + ".log")).read()[-1500:] + "\n")
    for d in sorted(glob.glob(os.path.join(out, "cfg4_pmc_*"))):
        if not os.path.isdir(d): continue
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(p)):
                k = (row.get("Kernel_Name", "?")[:90], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
            f.write("== PMC %s (sum over dispatches, n dispatches, per dispatch)\n" % os.path.basename(d))
            for (kn, cn), (v, n) in sorted(agg.items()):
                if "mlp_kernel" in kn:
                    f.write("%-92s %-28s %.6g  n=%d  avg=%.6g\n" % (kn, cn, v, n, v / max(n, 1)))
print(open(os.path.join(out, "summary.txt")).read()[:9000])
PY
