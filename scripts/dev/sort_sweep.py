"""Developer aid: device time of the segmented score sort, n lists of 5000, split sort (csrc/split_sort.hpp) against the
one-workgroup network and the counting kernel (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa

ctx = pa.Context(0)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for nseg in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256):
    s = rng.random(nseg * n)
    segs = np.arange(nseg + 1, dtype=np.uint32) * n
    row = []
    for mode, (split, rank) in (("split", (1000, 0)), ("network", (0, 0)), ("counting", (0, 1000))):
        ctx.set_option("split_sort_max", split)
        ctx.set_option("rank_sort_max", rank)
        best = 1e9
        for it in range(6):
            ctx.sort_scores(s, segs, descending=True)
            best = min(best, ctx.stats().last_sort_ms)
        row.append("%s %.1f us" % (mode, best * 1e3))
    print("lists %4d x %d: %s" % (nseg, n, " | ".join(row)))
