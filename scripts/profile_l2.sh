cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_l2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/scripts/dev/l2.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import sys, glob, csv, os
out = sys.argv[1]
lines = ["Round 3: squared-Euclidean recall (pg_recall_topk_l2), 100 M x 128, K = 5000 — rocprofv3 --kernel-trace --stats -- python3 scripts/dev/l2.py",
         "(N(0,1) rows: int8 screen, per-row test; normalised rows: int8 screen with per-block cutoffs; 1 / 32 / 64 / 128 / 256 queries per call, 4 calls each)", ""]
lines += [l.rstrip() for l in open(os.path.join(out, "log.txt")) if ("l2 nq" in l or "rows shadow" in l or "slice matches" in l)]
lines.append("")
for p in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if float(r["TotalDurationNs"]) > 2e5:
            lines.append("%-100s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
