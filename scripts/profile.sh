#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --latency-reqs 0 --no-extras --contexts 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_$name.log" 2>&1
done
# summaries
python3 - "$OUT" <<'PY'
import sys, os, csv, glob, collections
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        f.write("== kernel stats (%s)\n" % os.path.relpath(p, out))
        f.write(open(p).read() + "\n")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(p)):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
            f.write("== PMC %s (sum over dispatches, n dispatches)\n" % os.path.basename(d))
            for (kn, cn), (v, n) in sorted(agg.items()):
                if any(t in kn for t in ("scan_kernel", "screen_kernel", "rescore_kernel", "mlp_kernel", "dnn3_ws", "select", "sort_kernel")):
                    f.write("%-62s %-28s %.6g  n=%d  avg=%.6g\n" % (kn, cn, v, n, v / max(n, 1)))
print(open(os.path.join(out, "summary.txt")).read()[:6000])
PY
