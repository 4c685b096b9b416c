"""Developer probe: filtered views (pg_table_view_create) at the benchmark's table shape — build time, recall on the view by
batch size, per-request calls through a coalescer over the view from many threads."""
import sys, time, os, threading
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = 100_000_000, 128, 5000
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
feats = pa.Features(ctx, n)
col = np.random.default_rng(1).integers(0, 1_000_000, n).astype(np.int32)
feats.set_column("create_time", pa.F_I32, col)
for frac in (0.5, 0.1, 0.01):
    value = int(1_000_000 * (1 - frac))
    t0 = time.perf_counter(); v = t.view(feats, "create_time", ">=", value); build = (time.perf_counter() - t0) * 1e3
    line = f"view of {frac:5.2f} of the rows ({v.rows} rows): build {build:7.1f} ms;"
    for nq in (1, 16, 128, 256):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        for _ in range(6): v.recall_topk(q, k)           # statistics, shadows, threshold model
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); rows, sc, cnt = v.recall_topk(q, k); ts.append((time.perf_counter() - t0) * 1e3)
        line += f" nq={nq}: {sorted(ts)[3]:.2f} ms"
        if nq == 16:
            wr, ws, _ = t.recall_topk_where(feats, "create_time", ">=", value, q, k)
            assert np.array_equal(wr, rows) and np.array_equal(ws.view(np.uint32), sc.view(np.uint32))
    print(line, flush=True)
    co = pa.Coalescer(ctx, v, k, algos=[], max_wait_us=300)
    callers, per = 256, 12
    qs = o.synth_rows(o.SEED_QUERY, 100, callers, d)
    lat = [[] for _ in range(callers)]
    def run(i):
        for _ in range(per):
            t0 = time.perf_counter(); co.recall(qs[i]); lat[i].append((time.perf_counter() - t0) * 1e3)
    th = [threading.Thread(target=run, args=(i,)) for i in range(callers)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; wall = time.perf_counter() - t0
    allv = np.concatenate([np.array(x[2:]) for x in lat])
    print(f"   coalescer over the view, {callers} callers: {callers * per / wall:8.0f} requests/s, p50 {np.median(allv):.2f} ms p99 {np.percentile(allv, 99):.2f} ms", flush=True)
    co.destroy()
    v.destroy()
