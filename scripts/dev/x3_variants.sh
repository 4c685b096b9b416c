#!/bin/bash
# developer A/B of dnn3_x3_kernel build variants (each goes to libpairec_gpu_dev.so; the product library is untouched)
cd "$(dirname "$0")/../.."
for v in "$@"; do
  touch pairec_amd/csrc/rank_x3.hip
  make -C pairec_amd/csrc WS_EXTRA="$v" -j8 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  PG_LIB_VARIANT=dev X3_PRECS=2 python scripts/dev/x3_time.py 2>&1 | grep -E "wg 1 wave [04]|DNN3" | tail -3
done
