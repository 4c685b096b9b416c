cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lat1
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --latency-reqs 40 --no-extras --no-rank-shapes --contexts 1 --callers 0 --no-live-traffic > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import sys, csv, glob, os
rows=[]
for p in glob.glob(os.path.join(sys.argv[1],"**","*kernel_trace.csv"),recursive=True):
    rows+=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find last 'recall_init_kernel' start
idx=[i for i,r in enumerate(rows) if "recall_init_kernel" in r["Kernel_Name"]]
i0=idx[-2]; i1=idx[-1]          # the last but one single request of the latency leg
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%8.1f %8.1f %7.1f  %s" % (s,e,e-s,r["Kernel_Name"][:70]))
PY
