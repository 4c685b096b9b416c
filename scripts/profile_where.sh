cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_where
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/scripts/dev/where_prof.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import sys, glob, csv, os
out = sys.argv[1]
lines = ["Round 3: vector recall with a WhereClause (pg_recall_topk_where), 100 M x 128, K = 5000 — rocprofv3 --kernel-trace --stats -- python3 scripts/dev/where_prof.py",
         "(half / a tenth admitted: predicate in place on the screened plans; 4 % / 1 %: compact copy of the admitted rows + exact scan; 1 and 128 queries per call, 6 calls each;",
         " call times include the profiler's overhead)", ""]
lines += [l.rstrip() for l in open(os.path.join(out, "log.txt")) if l.startswith("where admitted")]
lines.append("")
for p in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if float(r["TotalDurationNs"]) > 2e5:
            lines.append("%-100s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
