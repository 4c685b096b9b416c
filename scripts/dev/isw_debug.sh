#!/bin/bash
cd "$(dirname "$0")/../.."
touch pairec_amd/csrc/rank_is.hip
make -C pairec_amd/csrc WS_EXTRA=-DPG_ISW_DEBUG -j8 > /dev/null 2>&1
# (built into libpairec_gpu_dev.so and loaded through PG_LIB_VARIANT=dev: the product library stays as it is)
for m in 1 2 3 4 5 6 7 0; do PG_LIB_VARIANT=dev python scripts/dev/isw_debug.py $m 2>&1 | tail -3; done
