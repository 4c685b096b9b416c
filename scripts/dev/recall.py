"""Developer probe: table fill + recall parity and scan throughput on one MI355X."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

big = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = pa.Context(0)
ok = True
for (n, d, k, nq) in [(1000, 64, 50, 1), (50_000, 128, 200, 3), (200_000, 128, 500, 32), (300_017, 64, 5000, 7)]:
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    ref = o.synth_rows(o.SEED_TABLE, 0, n, d)
    got = t.download(0, n)
    same = np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    rows, scores, cnt = t.recall_topk(q, k)
    orow, osc = o.recall_topk(ref, q, k)
    r_ok = np.array_equal(rows, orow); s_ok = np.array_equal(scores.view(np.uint32), osc.view(np.uint32))
    print(f"n={n} d={d} k={k} nq={nq}: fill_bitexact={same} rows_exact={r_ok} scores_bitexact={s_ok} rescans={ctx.stats().recall_rescans}")
    if not (r_ok and s_ok):
        bad = np.argwhere(rows != orow)[:5]
        print("  first mismatches", bad.tolist(), rows[tuple(bad[0])] if len(bad) else None, orow[tuple(bad[0])] if len(bad) else None)
    ok &= same and r_ok and s_ok
    t.destroy()
print("PARITY", "OK" if ok else "FAIL")
if big:
    n, d, k = big, 128, 5000
    t0 = time.time(); t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE); print(f"fill {n}x{d}: {time.time()-t0:.2f}s")
    for nq in (1, 32):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        for it in range(3):
            t0 = time.time(); rows, scores, cnt = t.recall_topk(q, k); dt = time.time() - t0
            ms, b = ctx.last_scan_kernel()
            st = ctx.stats()
            print(f"nq={nq} it={it}: wall {dt*1e3:.2f} ms, recall(dev) {st.last_recall_ms:.3f} ms, scan kernels {ms:.3f} ms -> {b/ms/1e9:.2f} TB/s... rescans={st.recall_rescans}")
    # spot-check top rows against oracle on a slice containing the winners
    sl = o.synth_rows(o.SEED_TABLE, int(rows[0, 0]), 1, d)
    print("top1 score", scores[0, 0], "oracle", o.dot_scores(sl, q[:1])[0, 0])
