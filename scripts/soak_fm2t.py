"""Soak: FM + two-tower rank on ragged request batches — the three item-side paths (field ids handed in, ids from feature
columns, materialised item records) against each other and against the oracle: request sizes 0..6000, 1..300 requests, field
ids at the vocabulary's ends and outside it (clamped), candidate rows with repeats and past the feature store's rows (column
defaults), both precisions.  F32: the three paths bit-identical, <= 3e-7 from the oracle; BF16: <= 1.5e-5.
Usage: soak_fm2t.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pa.Context(0)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
t_end = time.time() + seconds
cases = bad = 0
worst = {0: 0.0, 1: 0.0, 2: 0.0}
bf16_cases = bf16_over = 0
while time.time() < t_end:
    vocab = int(rng.choice([50, 3000, 200_000]))
    fw = o.Fm2tWeights(vocab=vocab, seed=o.SEED_WEIGHTS ^ int(rng.integers(0, 1000)))
    n_cat = int(rng.choice([500, 60_000]))
    ids = rng.integers(0, vocab, (n_cat, 8)).astype(np.int32)
    ids[rng.random((n_cat, 8)) < 0.02] = 0
    ids[rng.random((n_cat, 8)) < 0.02] = vocab - 1
    feats = pa.Features(ctx, n_cat)
    cols = ["f%d" % f for f in range(8)]
    for f, c in enumerate(cols):
        feats.set_column(c, pa.F_I32, np.ascontiguousarray(ids[:, f]))
    for prec in (0, 1, 2):                                   # 2 = PG_PREC_BF16X3, against the fp32 specification (round 5)
        m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, (pa.PREC_F32, pa.PREC_BF16, pa.PREC_BF16X3)[prec], pa.pack_fm2t(fw))
        ir = pa.ItemRows(m, feats, cols)
        for _ in range(3):
            R = int(rng.choice([1, 3, 64, 256, 300]))
            sizes = rng.choice([0, 1, 127, 128, 129, 1000, 6000], R, p=[.1, .1, .15, .15, .15, .3, .05])
            if sizes.sum() > 300_000:
                sizes = np.minimum(sizes, 1000)
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
            nI = int(off[-1])
            cand = rng.integers(0, n_cat, max(nI, 1)).astype(np.uint32)[:nI]
            if nI > 4:
                cand[:2] = (0, n_cat - 1)
                cand[2:4] = cand[0]
            users = o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 900)), R, 128)
            ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
            desc = dict(vocab=vocab, n_cat=n_cat, prec=prec, R=R, items=nI)
            try:
                if nI == 0:
                    cases += 1
                    continue
                a = m.rank_fm2t(users, ufids, ids[cand], off)
                b = m.rank_fm2t_rows(feats, cols, users, ufids, cand, off)
                c = ir.rank(users, ufids, cand, off)
            except Exception as ex:
                print("FAILED CASE", desc, repr(ex), flush=True)
                bad += 1
                cases += 1
                continue
            same = np.array_equal(bits(a), bits(b)) and (prec == 1 or np.array_equal(bits(a), bits(c)))   # (bf16: the record kernels sum the head in their own order)
            err = float(np.max(np.abs(a.astype(np.float64) - c.astype(np.float64)))) if prec else 0.0
            for r in sorted(set(int(x) for x in rng.integers(0, R, 3))):
                x, y = int(off[r]), int(off[r + 1])
                if y > x:
                    want = o.fm2t_forward(fw, prec & 1, users[r], ufids[r], ids[cand[x:y]])
                    err = max(err, float(np.max(np.abs(c[x:y].astype(np.float64) - want))), float(np.max(np.abs(a[x:y].astype(np.float64) - want))))
            worst[prec] = max(worst[prec], err)
            cases += 1
            if prec == 1:
                bf16_cases += 1
                bf16_over += err > 1e-5
            if prec == 2 and err > 2e-6:
                print("NOTE bf16x3 beyond its 2e-6 regression bar (inside north_star's 1e-5)", desc, "max abs err", err, flush=True)
            # (bf16x3: north_star's 1e-5 against the FP32 oracle; plain bf16: 1.5e-5 against the oracle that rounds where it rounds,
            #  its excursions beyond 1e-5 counted and reported)
            if not same or not (err <= (3e-7, 1.5e-5, 1e-5)[prec]):
                bad += 1
                print("MISMATCH", desc, "paths identical", same, "max abs err", err, flush=True)
        ir.destroy()
        m.destroy()
    feats.destroy()
print(f"soak_fm2t: {cases} batches, {bad} bad; worst |error| vs the oracle: fp32 mode {worst[0]:.2e}, bf16 mode {worst[1]:.2e}, bf16x3 mode {worst[2]:.2e} "
      f"(bf16x3 fails at north_star's 1e-5 against the fp32 oracle); plain bf16 exceeded 1e-5 against its MIRRORING oracle in "
      f"{bf16_over} of {bf16_cases} batches", flush=True)
sys.exit(1 if bad else 0)
