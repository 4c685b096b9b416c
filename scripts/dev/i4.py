"""Developer probe: the 4-bit small-batch screen against the int8 screen — identical results, time per recall."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pa.Context(0)
ctx.set_option("debug_scan", "1")
t = pa.Table(ctx, n, 128)
for kind in ("uniform", "gauss"):
    if kind == "uniform":
        t.fill_synthetic(o.SEED_TABLE)
    else:
        t.fill_gaussian(7, 0.05)
    for nq in (1, 2, 3, 4):
        q = o.synth_rows(o.SEED_QUERY, 10 * nq, nq, 128)
        res = {}
        for mode in ("i8", "i4"):
            ctx.set_option("no_screen_i4", "1" if mode == "i8" else "0")
            ctx.set_option("debug_scan", "1")
            t.recall_topk(q, 5000)
            ctx.set_option("debug_scan", "0")
            ts = []
            for it in range(5):
                t0 = time.time(); rows, sc, cnt = t.recall_topk(q, 5000); ts.append((time.time() - t0) * 1e3)
            st = ctx.stats()
            res[mode] = (rows, sc)
            print(f"{kind} nq={nq} {mode}: wall min {min(ts):.3f} ms, recall(dev) {st.last_recall_ms:.3f} ms, rescans={st.recall_rescans}", flush=True)
        same = np.array_equal(res["i8"][0], res["i4"][0]) and np.array_equal(res["i8"][1].view(np.uint32), res["i4"][1].view(np.uint32))
        print(f"  identical: {same}")
ctx.set_option("debug_scan", "1")
q = o.synth_rows(o.SEED_QUERY, 0, 1, 128)
t.recall_topk(q, 5000)
