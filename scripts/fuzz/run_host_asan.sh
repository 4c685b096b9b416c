#!/bin/bash
# ASAN + UBSan build of the host mirror (CPU; links the in-tree libpairec_gpu.so, no GPU call is made) and a mutation fuzz of its
# parsers.  Usage: scripts/fuzz/run_host_asan.sh [seed] [count]
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=/tmp/pairec_asan
mkdir -p $OUT
cd $REPO/pairec_amd/host
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $OUT/libpairec_host_asan.so \
    pairec_host.cpp loadgen.cpp ingest.cpp -lpthread -L.. -lpairec_gpu -Wl,-rpath,$REPO/pairec_amd
cd $REPO
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 PH_ASAN_LIB=$OUT/libpairec_host_asan.so \
    python3 scripts/fuzz/fuzz_host.py "${1:-1}" "${2:-20000}"
