#!/bin/bash
# Developer aid (GPU box): what the package draws and clocks at while one kernel runs back to back — a long loop of the split-bf16 rank stage,
# then of 256-query recall passes — sampled with rocm-smi from a second process.  Usage: scripts/dev/power_probe.sh
cd "$(dirname "$0")/../.."
probe() {
  for i in 1 2 3 4 5 6; do
    sleep 0.7
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr '\n' ';'
    echo
  done
}
echo "== idle"; probe | head -2
echo "== dnn3_x3_kernel loop"
python scripts/dev/x3_time.py 100000000 loop > /tmp/x3loop.log 2>&1 &
P=$!
sleep 6
probe
wait $P
tail -1 /tmp/x3loop.log
echo "== 256-query recall passes"
python scripts/dev/i4m_prof.py 256 100000000 loop > /tmp/rloop.log 2>&1 &
P=$!
sleep 6
probe
wait $P
tail -1 /tmp/rloop.log
