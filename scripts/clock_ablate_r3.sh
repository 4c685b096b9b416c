#!/bin/bash
# (round 5: such a build is libpairec_gpu_dev.so — export PG_LIB_VARIANT=dev for the runs below)
# developer aid: sustained shader clock of the 256-query screen's ablation variants (needs a build with
# SCAN_EXTRA=-DPG_SCAN_VARIANTS): GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the largest screen_kernel dispatches
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/clk_r3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PG_NO_PREDICT=1 PG_NO_REFINE=1
for v in ${VARS:-0 1 2 5}; do
  export PG_SCREEN_VAR=$v
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/v$v" -o p -- python3 "$REPO/bench.py" --steps 4 --warmup 2 --calibrate 0 --no-cpu-baseline --latency-reqs 0 --no-extras --no-rank-shapes --callers 0 --contexts 1 > "$OUT/v$v.log" 2>&1
  python3 - "$OUT/v$v" $v <<'PY'
import sys, csv, glob, os, collections
d, v = sys.argv[1], sys.argv[2]
dur = {}
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "screen_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cnt = collections.defaultdict(dict)
for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "screen_kernel" in r["Kernel_Name"]:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
big = sorted(dur.items(), key=lambda x: -x[1])[:4]
for did, us in big:
    c = cnt.get(did, {})
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8
    print("var %s: dispatch %s %.0f us, clock %.2f GHz, mfma busy %.2f" % (v, did, us, g / us / 1e3, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0))
PY
done
