#!/bin/bash
# Developer aid (GPU box): what the package draws and clocks at while ONE kernel runs back to back — a long loop of the split-bf16 rank
# stage, of 256-query recall passes, of 32-query passes (4-bit shadow), of cfg 4's rank kernel — sampled with rocm-smi from a second
# process.  (bench.py's `power` object does the same in-process for the headline.)  Usage: scripts/dev/power_probe.sh
cd "$(dirname "$0")/../.."
probe() {
  for i in 1 2 3 4; do
    sleep 0.7
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '
    echo
  done
}
run() {   # name, script + args
  echo "== $1"
  shift
  python "$@" > /tmp/pp.log 2>&1 &
  P=$!
  sleep 7
  probe
  wait $P
  tail -1 /tmp/pp.log | cut -c1-200
}
echo "== idle"; probe | sed -n 1,2p
run "dnn3_x3_kernel (rank stage alone)" scripts/dev/x3_time.py 100000000 loop
run "256-query recall passes (int8 shadow)" scripts/dev/i4m_prof.py 256 100000000 loop
run "64-query recall passes (4-bit shadow, two query blocks)" scripts/dev/i4m_prof.py 64 100000000 loop
run "32-query recall passes (4-bit shadow)" scripts/dev/i4m_prof.py 32 100000000 loop
run "1-query recall passes (4-bit shadow)" scripts/dev/i4m_prof.py 1 100000000 loop
run "cfg 4 rank kernel (fm2t_isw_kernel)" scripts/dev/cfg4_prof.py random loop
