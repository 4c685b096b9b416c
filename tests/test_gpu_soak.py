"""A short run of scripts/soak_coalescer.py: random bursts of concurrent callers (recall / rank / recommend, random
coalescer settings) against reference answers from the batch API — with the pilot plan's threshold refinement forced on
(and the 4-bit small-batch screen) for the 3 M-row table, so that the verification / partial re-run path is in the loop too.
Looks for races."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_coalescer_soak_short():
    env = dict(os.environ, PG_REFINE_MIN_ROWS="0", PG_I4_MIN_ROWS="0")      # refinement and the 4-bit screen on the 3 M-row table
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_coalescer.py"), "8"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "bad 0" in r.stdout, tail
