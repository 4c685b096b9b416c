#!/bin/bash
# AddressSanitizer + UBSan over the host C++ layer, CPU box only: `make -C pairec_amd/host asan` builds
# pairec_amd/libpairec_host_asan.so from the product's sources; the CPU mirror / normalizer / antlr tests run against it, then
# scripts/host_fuzz.py drives mutated inputs through every parser for FUZZ_SECONDS (default 60).  Exit 0 = no report.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
make -s -C pairec_amd/csrc
make -s -C pairec_amd/host asan
export PH_HOST_LIB="$REPO/pairec_amd/libpairec_host_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="halt_on_error=1:abort_on_error=1:print_stacktrace=1"
python3 -m pytest tests/test_host_mirror.py tests/test_feature_normalizer.py tests/test_expr_antlr.py -x -q -m "not gpu" -p no:cacheprovider
python3 scripts/host_fuzz.py "${FUZZ_SECONDS:-60}" "${1:-1}"
echo "host_asan: clean"
