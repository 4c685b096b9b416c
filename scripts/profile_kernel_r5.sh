#!/bin/bash
# Round-5 profile of ONE kernel: a kernel trace and one rocprofv3 --pmc pass per counter group, the program directly behind
# `--` (never a shell: the profiler's preloaded library has initialised the GPU).
# Usage: scripts/profile_kernel_r5.sh <tag> <kernel-name substring> <python script> [args...]
set -u
TAG=$1; KSUB=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
PROG=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$PROG" "$@" > "$OUT/trace.log" 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$PROG" "$@" > "$OUT/pmc_$name.log" 2>&1
done
python3 - "$OUT" "$KSUB" <<'PY'
import sys, os, csv, glob, collections
out, ksub = sys.argv[1], sys.argv[2]
lines = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        lines.append("%-90s calls %5s avg_us %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            kn = r.get("Kernel_Name", "?")
            if ksub in kn:
                by[(kn[:60], r.get("Counter_Name", "?"))].append(float(r.get("Counter_Value", 0) or 0))
        lines.append("== %s (mean per dispatch)" % os.path.basename(d))
        for (kn, cn), v in sorted(by.items()):
            lines.append("%-62s %-26s %.6g" % (kn, cn, sum(v) / len(v)))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
