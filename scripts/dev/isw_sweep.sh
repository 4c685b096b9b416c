#!/bin/bash
# developer experiment: waves per CU of fm2t_isw_kernel
cd "$(dirname "$0")/../.."
for w in 8; do
  touch pairec_amd/csrc/rank_is.hip
  make -C pairec_amd/csrc WS_EXTRA=-DPG_ISW_WAVES=$w -j8 > /dev/null 2>&1
  echo "waves $w"
  for i in 1 2; do python scripts/dev/cfg4_prof.py random 2>&1 | tail -1; done
  python scripts/dev/cfg4_prof.py row0 2>&1 | tail -1
done
