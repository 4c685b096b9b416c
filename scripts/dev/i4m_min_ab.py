"""Developer aid: one- and two-query passes on the vector 4-bit screen (screen4_kernel) against the matrix-pipe one
(screen4m_kernel, i4m_min_queries lowered), same box, scan-stage ms (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

ctx = pa.Context(0)
t = pa.Table(ctx, 100_000_000, 128)
if len(sys.argv) > 1 and sys.argv[1] == "gauss":
    t.fill_gaussian(o.SEED_TABLE, 1.0)
else:
    t.fill_synthetic(o.SEED_TABLE)
K = 5000
q = o.synth_rows(o.SEED_QUERY, 0, 4096, 128)
d_q = ctx.to_device(q)
d_rows = ctx.malloc(256 * K * 8)
d_sc = ctx.malloc(256 * K * 4)
for i in range(12):
    t.recall_topk_dev(d_q + (i % 8) * 256 * 128 * 4, 256, K, d_rows, d_sc)
for rnd in range(2):
    for mn in (3, 1):
        ctx.set_option("i4m_min_queries", mn)
        for R in (1, 2, 3, 4):
            ms = []
            for it in range(8):
                t.recall_topk_dev(d_q + ((it * 7 + 3) % 15) * 256 * 128 * 4, R, K, d_rows, d_sc)
                ctx.synchronize()
                ms.append(ctx.last_scan_kernel()[0])
            print("i4m_min_queries %d  R %d  scan ms min %.3f median %.3f" % (mn, R, min(ms[2:]), float(np.median(ms[2:]))))
