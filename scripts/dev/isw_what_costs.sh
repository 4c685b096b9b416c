#!/bin/bash
# developer experiment: what the towers cost the gather in fm2t_isw_kernel — without their LDS weight reads, without their
# MFMAs (results wrong in both), against the full kernel and the gather alone
cd "$(dirname "$0")/../.."
# (every variant goes to libpairec_gpu_dev.so — csrc/Makefile — and is loaded through PG_LIB_VARIANT=dev; the product
# library is not rebuilt or replaced)
for v in "-DPG_ISW_NO_LDSW" "-DPG_ISW_NO_MFMA" "-DPG_ISW_NO_MFMA -DPG_ISW_NO_LDSW" "-DPG_ISW_GATHER_ONLY" "-DPG_ISW_PRODUCT_AS_DEV"; do
touch pairec_amd/csrc/rank_is.hip
make -C pairec_amd/csrc WS_EXTRA="$v" -j8 > /dev/null 2>&1
echo "variant [$v]"
for i in 1 2 3; do PG_LIB_VARIANT=dev python scripts/dev/cfg4_prof.py random 2>&1 | tail -1; done
PG_LIB_VARIANT=dev python scripts/dev/cfg4_prof.py row0 2>&1 | tail -1
done
