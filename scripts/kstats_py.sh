#!/bin/bash
# kernel-trace stats of an arbitrary python script: scripts/kstats_py.sh <tag> <script.py> [args]
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/ks_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$REPO/$1" "${@:2}" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, os
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
