#!/bin/bash
# developer A/B of dnn3_t3_kernel build variants (each goes to libpairec_gpu_dev.so; the product library is untouched)
cd "$(dirname "$0")/../.."
for v in "$@"; do
  touch pairec_amd/csrc/rank_t3.hip
  make -C pairec_amd/csrc WS_EXTRA="$v" -j8 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  PG_LIB_VARIANT=dev T3_KNOBS=1,1 python scripts/dev/t3_time.py 2>&1 | grep -E "wave  [048]:|^t3" | tail -5
done
