#!/bin/bash
# Round-6 profile of one program in its steady state: rocprofv3 --kernel-trace --stats, then PMC groups in runs of their own
# (no trace domain with --pmc; the program directly after `--`).  GPU box, via gpurun.
# Usage: scripts/profile_r6.sh <tag> <kernel substrings, comma-separated> <python script + args ...>
#   scripts/profile_r6.sh sweep_32 screen4m_kernel,rescreen8_kernel,rescore_kernel scripts/dev/i4m_prof.py 32
#   scripts/profile_r6.sh headline screen_kernel,screen_decode,rescore_kernel,dnn3_x3 bench.py --steps 6 --warmup 2 --no-cpu-baseline \
#       --latency-reqs 0 --no-extras --no-rank-shapes --no-f32-leg --no-batch-sweep --contexts 1
# The summary lists the LAST `TAIL` dispatches of each named kernel (the first batches of a run are calibration on the pilot plan).
set -u
TAG=$1; KERNELS=$2; shift 2
TAIL=${TAIL:-6}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_r6_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp OMP_WAIT_POLICY=PASSIVE
PROG="$REPO/$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$PROG" "$@" > "$OUT/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$PROG" "$@" > "$OUT/pmc_$name.log" 2>&1
done
python3 - "$OUT" "$TAIL" "$KERNELS" "$TAG" "$(basename $PROG) $*" <<'PY'
import sys, os, csv, glob, collections, json
out, tail, kernels, tag, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3].split(","), sys.argv[4], sys.argv[5]
lines = ["Round 6, %s: rocprofv3 --kernel-trace --stats / --pmc <group> -- python3 %s (scripts/profile_r6.sh); per kernel the last %d "
         "dispatches of the run = steady state; fetch_bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half the bytes of 16 B/lane "
         "reads, MI355X_MICROARCH.md), write_bytes = WRITE_SIZE (KB) x 1024; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
         "GRBM_GUI_ACTIVE / 8 XCDs)" % (tag, cmd, tail)]
res = {"_how": lines[0]}
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats, whole run incl. warm-up (%s)" % os.path.relpath(p, out))
    lines.append(open(p).read())
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        kn = r["Kernel_Name"]
        if any(k in kn for k in kernels):
            by[kn[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines.append("== last %d dispatches per kernel (us), kernel trace" % tail)
    for kn, v in sorted(by.items()):
        t = v[-tail:]
        lines.append("%-92s n=%4d  last: %s  mean %.1f" % (kn, len(v), " ".join("%.1f" % x for x in t), sum(t) / len(t)))
        res.setdefault("kernel_us", {})[kn] = {"dispatches": len(v), "last_mean_us": sum(t) / len(t)}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            kn = r.get("Kernel_Name", "?")
            if any(k in kn for k in kernels):
                by[(kn[:90], r.get("Counter_Name", "?"))].append((int(r.get("Dispatch_Id", 0)), float(r.get("Counter_Value", 0) or 0)))
        lines.append("== PMC %s: mean over the last %d dispatches" % (os.path.basename(d), tail))
        for (kn, cn), v in sorted(by.items()):
            v.sort()
            t = [x[1] for x in v[-tail:]]
            m = sum(t) / len(t)
            lines.append("%-92s %-28s %.6g" % (kn, cn, m))
            res.setdefault("pmc", {}).setdefault(kn, {})[cn] = m
for kn, c in res.get("pmc", {}).items():
    if "FETCH_SIZE" in c:
        c["fetch_bytes"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        c["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    lines.append("== derived, %s: %s" % (kn, {k: (round(v, 4) if isinstance(v, float) and v < 10 else v) for k, v in c.items() if k in ("fetch_bytes", "write_bytes", "mfma_busy_frac")}))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("\n".join(lines)[-5000:])
PY
