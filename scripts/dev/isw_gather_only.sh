#!/bin/bash
# developer experiment: fm2t_isw_kernel without its towers (-DPG_ISW_GATHER_ONLY) — what the gather + FM sums alone reach
cd "$(dirname "$0")/../.."
for v in "-DPG_ISW_GATHER_ONLY" ""; do
touch pairec_amd/csrc/rank_is.hip
make -C pairec_amd/csrc WS_EXTRA="$v" -j8 > /dev/null 2>&1
echo "variant [$v]"
for i in 1 2 3; do python scripts/dev/cfg4_prof.py random 2>&1 | tail -1; done
python scripts/dev/cfg4_prof.py row0 2>&1 | tail -1
done
