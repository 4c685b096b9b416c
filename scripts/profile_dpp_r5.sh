#!/bin/bash
# Round-5 PMC passes of the batched DPP stage (256 requests x 500 candidates x 129 columns -> 100 picks): where
# dpp_kernel_matrix_kernel's time goes.  Program directly behind `--`, one counter group per run, no trace domains with --pmc.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_dpp_r5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/scripts/dev/dpp_batch.py" > "$OUT/trace.log" 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$REPO/scripts/dev/dpp_batch.py" > "$OUT/pmc_$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import sys, os, csv, glob, collections
out = sys.argv[1]
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
            if "dpp_" in kn:
                by[(kn[:60], r.get("Counter_Name", "?"))].append(float(r.get("Counter_Value", 0) or 0))
        lines.append("== %s (mean per dispatch)" % os.path.basename(d))
        for (kn, cn), v in sorted(by.items()):
            lines.append("%-62s %-26s %.6g" % (kn, cn, sum(v) / len(v)))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
