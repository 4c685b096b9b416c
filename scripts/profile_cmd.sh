#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of an arbitrary python command.
# Usage: scripts/profile_cmd.sh <tag> <kernel-name-filter> <script.py> [args...]
set -u
TAG=$1; FILT=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$@" > "$OUT/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$@" > "$OUT/pmc_$name.log" 2>&1
done
python3 - "$OUT" "$FILT" <<'PY'
import sys, os, csv, glob, collections
out, filt = sys.argv[1], sys.argv[2].split(",")
with open(os.path.join(out, "summary.txt"), "w") as f:
    for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        f.write("== kernel stats (%s)\n" % os.path.relpath(p, out))
        f.write(open(p).read() + "\n")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(p)):
                k = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
            f.write("== PMC %s (sum over dispatches, n dispatches)\n" % os.path.basename(d))
            for (kn, cn), (v, n) in sorted(agg.items()):
                if any(t in kn for t in filt):
                    f.write("%-72s %-28s %.6g  n=%d  avg=%.6g\n" % (kn, cn, v, n, v / max(n, 1)))
print(open(os.path.join(out, "summary.txt")).read()[:7000])
PY
