#!/bin/bash
# Every randomised / adversarial soak, S seconds each (default 60), on the GPU box:
#   gpurun --timeout 1500 -- 'bash scripts/run_soaks.sh 60 [seed]'
# Exit status 1 if any of them reports a mismatch.  DESIGN.md 3 lists what each one covers.
S=${1:-60}
SEED=${2:-1}
cd "$(dirname "$0")/.."
rc=0
for s in soak_adversarial soak_where soak_pipeline_adversarial soak_group soak_scene soak_coalescer soak_records soak_l2 soak_rank soak_fm2t soak_rerank soak_expr soak_antlr_rewrite; do
  echo "== $s ($S s)"
  timeout $((S * 4 + 300)) python3 scripts/$s.py "$S" "$SEED" 2>&1 | grep -a "MISMATCH\|FAILED\|^soak\|^rounds\|fault\|Traceback" | tail -4 || true
  [ "${PIPESTATUS[0]}" = "0" ] || rc=1
done
echo "== soak_blobs, soak_misuse"
python3 scripts/soak_blobs.py 2000 "$SEED" 2>&1 | tail -1 || rc=1
python3 scripts/soak_misuse.py 2>&1 | tail -1 || rc=1
exit $rc
