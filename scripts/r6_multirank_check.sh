#!/bin/bash
# Round 6 (as round 5): developer check of the N > 1 paths on ONE GPU (PG_BENCH_SHARE_GPU=1: every rank on cuda:0, gloo with host-staged
# payloads) — the preflight, the replica headline with its shard / group sub-objects, and the shard / group modes.
# Correctness of the launch paths, not a measurement.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_multirank_dev
export PG_BENCH_SHARE_GPU=1
COMMON="--rows 3000000 --steps 3 --warmup 1 --calibrate 2 --no-cpu-baseline"
for spec in "replica 2" "shard 2" "group 4" "router 2"; do
  set -- $spec
  timeout 900 python bench.py --gpus $2 --mode $1 $COMMON > gpurun_out/r6_multirank_dev/$1_$2.json 2> gpurun_out/r6_multirank_dev/$1_$2.err
  echo "mode $1 x$2: rc $?"
  python - "$1" "$2" <<'PY'
import json, sys
mode, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("gpurun_out/r6_multirank_dev/%s_%s.json" % (mode, n)).read().strip().splitlines()[-1])
except Exception as e:
    print("  no JSON line:", e); sys.exit(0)
pf = d.get("preflight")
print("  value %s  n_gpus %s  dev_mode %s" % (d.get("value"), d.get("n_gpus"), bool(d.get("dev_mode"))))
print("  preflight:", json.dumps(pf)[:400])
print("  workload:", d["config"]["workload"][:120])
for k in ("shard", "group"):
    if k in d:
        x = dict(d[k]); c = x.pop("config", {})
        print("  %s: %s | %s" % (k, json.dumps(x), c.get("workload", "")[:90]))
print("  spot:", json.dumps(d.get("oracle_spot_check"))[:300])
PY
  tail -3 gpurun_out/r6_multirank_dev/$1_$2.err
done
