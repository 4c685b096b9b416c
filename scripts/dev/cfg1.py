"""Developer probe: cfg 1 (1M x 64, top-200, one request per call) recall latency under different pilot fractions."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
for (n, d, k) in ((1_000_000, 64, 200), (1_000_000, 128, 200), (4_000_000, 128, 1000), (300_000, 64, 200)):
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    q = o.synth_rows(o.SEED_QUERY, 0, 1, d)
    base = None
    for frac in ("0", "0.5", "0.25", "0.125", "0.0625"):
        ctx.set_option("pilot_fraction", frac)
        rows, sc, _ = t.recall_topk(q, k)
        ts = []
        for _ in range(20):
            t0 = time.time(); t.recall_topk(q, k); ts.append((time.time() - t0) * 1e3)
        st = ctx.stats()
        if base is None:
            base = (rows, sc)
        same = np.array_equal(rows, base[0]) and np.array_equal(sc.view(np.uint32), base[1].view(np.uint32))
        print(f"n={n} d={d} k={k} pilot_fraction={frac}: wall p50 {np.median(ts):.3f} ms, dev {st.last_recall_ms:.3f} ms, launches {ctx.last_scan_kernel()} same={same} rescans={st.recall_rescans}", flush=True)
    ctx.set_option("pilot_fraction", "0")
    t.destroy()
