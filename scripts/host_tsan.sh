#!/bin/bash
# ThreadSanitizer over the host C++ layer's shared state, CPU box only: `make -C pairec_amd/host tsan` builds
# pairec_amd/libpairec_host_tsan.so; scripts/host_threads.py then calls the registry / RandomNormalizer / IdDict / recconf entry
# points from eight threads at once.  Exit 0 = no race report.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
make -s -C pairec_amd/csrc
make -s -C pairec_amd/host tsan
export PH_HOST_LIB="$REPO/pairec_amd/libpairec_host_tsan.so"
export LD_PRELOAD="$(gcc -print-file-name=libtsan.so)"
export TSAN_OPTIONS="halt_on_error=1:abort_on_error=1:second_deadlock_stack=1:report_signal_unsafe=0:suppressions=$REPO/scripts/host_tsan.supp"
python3 scripts/host_threads.py "${1:-8}" "${2:-10}"
echo "host_tsan: clean"
