"""Soak of the filtered recalls: random tables (rows, dim, norm profile, duplicates), K, batch sizes, operators, selectivities and
metrics — pg_recall_topk_where (in place and compact routes) and recalls over pg_table_view_create views against the oracle run
on the admitted rows alone (ids, order, score bits).  Usage: soak_where.py [seconds] [seed]"""
import sys, time, os
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pa.Context(0)
if os.environ.get("SOAK_SMALL_TABLE_PLANS"):            # the 4-bit screen, the threshold model and the refinement on these tables and views
    for opt, val in (("i4_min_rows", 1024), ("predict_min_rows", 0), ("refine_min_rows", 1024)):
        ctx.set_option(opt, val)
OPS = {">": np.greater, ">=": np.greater_equal, "<": np.less, "<=": np.less_equal, "==": np.equal, "!=": np.not_equal}
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
t_end = time.time() + budget
n_tables = n_cases = n_checked = 0
while time.time() < t_end:
    n = int(rng.choice([3000, 50_000, 400_000, 1_500_000, 3_000_000]))
    d = int(rng.choice([64, 128, 128]))
    tab = o.synth_rows(o.SEED_TABLE, int(rng.integers(0, 1 << 30)), n, d)
    prof = rng.integers(0, 5)
    if prof >= 3:                                       # score order ascending / descending in the row index (for queries near v)
        v_ = rng.standard_normal(d).astype(np.float32)
        v_ /= np.linalg.norm(v_)
        ramp = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
        tab = (v_[None] * (ramp if prof == 3 else ramp[::-1]) + 0.002 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    if prof == 1:
        tab = (tab * rng.uniform(0.6, 1.5, (n, 1)).astype(np.float32)).astype(np.float32)
    elif prof == 2:
        tab = rng.standard_normal((n, d)).astype(np.float32)
    if n > 5000:
        a, b = rng.integers(0, n - 200, 2)
        tab[a:a + 50] = tab[b:b + 50]
    off = int(rng.choice([0, 0, 1 << 22]))
    t = pa.Table(ctx, n, d, row_offset=off)
    t.upload(tab)
    feats = pa.Features(ctx, n)
    card = int(rng.choice([2, 10, 100, 10_000, 1_000_000]))
    col32 = rng.integers(0, card, n).astype(np.int32)
    col64 = (rng.integers(0, card, n).astype(np.int64) - (1 << 40))
    if rng.random() < 0.4:                              # a column that grows with the row (create_time of a table in insertion order):
        col32 = (np.arange(n) // max(n // card, 1)).astype(np.int32)          # filters admit contiguous row ranges
        col64 = col32.astype(np.int64) - (1 << 40)
    feats.set_column("c32", pa.F_I32, col32)
    feats.set_column("c64", pa.F_I64, col64)
    n_tables += 1
    for _ in range(6):
        use64 = bool(rng.integers(0, 2))
        vals = col64 if use64 else col32
        op = str(rng.choice(list(OPS)))
        value = int(rng.choice(vals)) if rng.random() < 0.7 else int(rng.integers(int(vals.min()) - 1, int(vals.max()) + 2))
        mask = OPS[op](vals, value)
        idx = np.nonzero(mask)[0]
        k = int(rng.choice([1, 10, 200, 1000, 2000]))
        nq = int(rng.choice([1, 2, 4, 5, 33, 64, 128, 200, 256]))
        if d == 64 or d > 128:
            nq = min(nq, 200)
        l2 = bool(rng.integers(0, 2))
        q = (o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 1 << 20)), nq, d) * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
        if prof >= 3 and rng.random() < 0.7:
            q = (v_[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
        if rng.random() < 0.3 and idx.size:
            q[0] = tab[idx[0]]                                     # a query equal to an admitted row
        compact_off = rng.random() < 0.25
        if compact_off:
            ctx.set_option("where_compact_max_rows", 0)
        try:
            rows, sc, cnt = t.recall_topk_where(feats, "c64" if use64 else "c32", op, value, q, k, l2=l2)
        except Exception:
            print("FAILED CASE:", dict(n=n, d=d, prof=int(prof), off=off, card=card, use64=use64, op=op, value=value, admitted=int(idx.size), k=k, nq=nq, l2=l2,
                                        compact_off=compact_off), flush=True)
            raise
        if compact_off:
            ctx.set_option("where_compact_max_rows", 8 << 20)
        m = min(k, idx.size)
        assert cnt.tolist() == [m] * nq, (n, d, op, value, nq, k, l2, cnt[:4], m)
        sel = sorted(set(int(x) for x in rng.integers(0, nq, 3)) | {0})
        if m:
            orow, osc = (o.recall_topk_l2 if l2 else o.recall_topk)(tab[idx], q[sel], k)
            want_rows = (idx[orow[:, :m].astype(np.int64)] + off).astype(np.uint64)
            assert np.array_equal(rows[sel][:, :m], want_rows), ("where rows", n, d, op, value, nq, k, l2, compact_off)
            assert np.array_equal(bits(sc[sel][:, :m]), bits(osc[:, :m])), ("where scores", n, d, op, value, nq, k, l2, compact_off)
            n_checked += len(sel)
        assert np.all(rows[:, m:] == np.uint64(0xFFFFFFFFFFFFFFFF))
        if idx.size and rng.random() < 0.5:
            v = t.view(feats, "c64" if use64 else "c32", op, value)
            vr, vs, vc = (v.recall_topk_l2 if l2 else v.recall_topk)(q, k)
            assert np.array_equal(vr, rows) and np.array_equal(bits(vs[:, :m]), bits(sc[:, :m])) and vc.tolist() == cnt.tolist(), \
                ("view", n, d, op, value, nq, k, l2)
            v.destroy()
        n_cases += 1
    feats.destroy()
    t.destroy()
print(f"soak_where: {n_tables} tables, {n_cases} filtered recalls ({n_checked} query answers against the oracle, the rest view == per-call): all identical", flush=True)
