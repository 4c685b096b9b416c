"""Developer probe: where do the rare slow calls of a single-request loop come from?  Logs every call slower than twice the
median with its time offset, for (a) a 1-query recall, (b) a tiny table gather (launch + sync only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = 100_000_000, 128, 5000
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
q = o.synth_rows(o.SEED_QUERY, 0, 1, d)
rows = np.arange(64, dtype=np.uint32)
def loop(name, f, reps):
    for _ in range(20): f()
    ts, at = [], []
    t00 = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); at.append((t0 - t00) * 1e3)
    ts = np.array(ts); med = float(np.median(ts))
    slow = [(i, round(at[i], 1), round(float(ts[i]), 2)) for i in range(reps) if ts[i] > 2 * med]
    print(f"{name}: median {med:.3f} ms p99 {np.percentile(ts, 99):.3f} max {ts.max():.2f}; slow calls (index, at ms, took ms): {slow[:40]}", flush=True)
loop("gather 64 rows", lambda: t.gather(rows), 3000)
loop("recall 1 query", lambda: t.recall_topk(q, k), 1500)
loop("gather 64 rows", lambda: t.gather(rows), 3000)
loop("recall 1 query", lambda: t.recall_topk(q, k), 1500)
