"""Developer aid (GPU box): recalls on a clustered table (pg_table_fill_mixture) with the plans' own debug output.
   python scripts/dev/clustered.py [rows] [centres] [sigma] [nq] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
C_ = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 256
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 8
K = 5000
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
t.fill_mixture(0x5EED0007, C_, sigma)
print("screen_info", t.screen_info())
d_rows = ctx.malloc(256 * K * 8)
d_sc = ctx.malloc(256 * K * 4)
ctx.set_option("debug_scan", "1" if os.environ.get("PG_DBG") else "0")
for it in range(reps):
    q = o.synth_mixture_rows(0x5EED0007, 1000 * it, nq, 128, C_, sigma, stream=1)
    d_q = ctx.to_device(q)
    s0 = ctx.stats()
    t0 = time.perf_counter()
    t.recall_topk_dev(d_q, nq, K, d_rows, d_sc)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    s1 = ctx.stats()
    ms, by = ctx.last_scan_kernel()
    nqd = max(s1.recall_suspect_queries - s0.recall_suspect_queries, 1)
    print("it %d: wall %.2f ms scan %.2f ms bytes %.2f GB | suspects/answer %.2f rescored/answer %.2f rescans %d overflows %d predicted %d" % (
        it, wall, ms, by / 1e9, (s1.recall_suspects - s0.recall_suspects) / nqd / K, (s1.recall_rescored - s0.recall_rescored) / nqd / K,
        s1.recall_rescans - s0.recall_rescans, s1.recall_screen_overflows - s0.recall_screen_overflows, s1.recall_predicted - s0.recall_predicted), flush=True)
    ctx.free(d_q)
if os.environ.get("PG_CHECK"):
    tab = t.download(0, rows)
    q = o.synth_mixture_rows(0x5EED0007, 7, 8, 128, C_, sigma, stream=1)
    r, s, _ = t.recall_topk(q, K)
    orow, osc = o.recall_topk(tab, q, K)
    print("exact:", np.array_equal(r, orow) and np.array_equal(s.view(np.uint32), osc.view(np.uint32)))
