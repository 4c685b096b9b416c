"""Soak: the DNN3 rank stage (every hidden shape: the register-stationary, weights-stationary and streamed-weights bf16 kernels,
the fp32 kernel) on ragged request batches — request sizes 0..6000 incl. empty ones, 1..300 requests, candidates with repeats
and the table's first / last rows, tables of ordinary, zero and large rows, dim 64 and 128 — against the oracle's forward
(fp32 mode <= 2e-7; bf16 mode <= 1.5e-5 absolute on the sigmoid output for inputs of at most unit norm (1e-5 in all but ~1 of 10 000 batches), DESIGN.md 5.2 — the
bf16 mode's error is the accumulation order's and grows with the pre-activations: rows x users scaled up to 24 x are held to 1e-4).  Also the segmented sort on random
segments (empty, 1, long; NaN / inf / ties / signed zeros) against the oracle's order.
Usage: soak_rank.py [seconds] [seed]"""
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
SHAPES = ((128, 128), (256, 128), (256, 256), (512, 256), (1024, 512))
t_end = time.time() + seconds
cases = bad = 0
worst = {0: 0.0, 1: 0.0, 2: 0.0}
bf16_cases = bf16_over = 0
while time.time() < t_end:
    d_item = int(rng.choice([64, 128]))
    n = int(rng.choice([300, 50_000, 400_000]))
    kind = int(rng.integers(0, 3))
    tab = o.synth_rows(o.SEED_TABLE, int(rng.integers(0, 1 << 20)), n, d_item)
    if kind == 1:
        tab[rng.random(n) < 0.3] = 0.0
    elif kind == 2:
        tab *= rng.uniform(0.1, 8.0, (n, 1)).astype(np.float32)
    t = pa.Table(ctx, n, d_item)
    t.upload(tab)
    for _ in range(4):
        h1, h2 = SHAPES[int(rng.integers(0, len(SHAPES)))]
        prec = int(rng.integers(0, 3))                        # 2 = PG_PREC_BF16X3: compared with the fp32 specification (round 5)
        w = o.Dnn3Weights(d_user=128, d_item=d_item, h1=h1, h2=h2, seed=o.SEED_WEIGHTS ^ int(rng.integers(0, 1000)))
        m = pa.RankModel(ctx, pa.MODEL_DNN3, (pa.PREC_F32, pa.PREC_BF16, pa.PREC_BF16X3)[prec], pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
        R = int(rng.choice([1, 2, 7, 64, 256, 300]))
        sizes = rng.choice([0, 1, 63, 64, 65, 100, 129, 1000, 6000], R, p=[.1, .1, .1, .1, .1, .2, .1, .15, .05])
        if sizes.sum() > 400_000:
            sizes = np.minimum(sizes, 1000)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        nI = int(off[-1])
        cand = rng.integers(0, n, max(nI, 1)).astype(np.uint32)[:nI]
        if nI > 4:
            cand[:2] = (0, n - 1)
            cand[2:4] = cand[0]
        uscale = float(rng.uniform(0.2, 3.0))
        users = (o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 900)), R, 128) * np.float32(uscale)).astype(np.float32)
        desc = dict(n=n, d_item=d_item, kind=kind, h1=h1, h2=h2, prec=prec, R=R, items=nI, uscale=round(uscale, 2))
        try:
            got = m.rank_dnn3(t, users, cand, off) if nI else np.zeros(0, np.float32)
        except Exception as ex:
            print("FAILED CASE", desc, repr(ex), flush=True)
            bad += 1
            cases += 1
            m.destroy()
            continue
        # the oracle on a sample of requests
        err = 0.0
        for r in sorted(set(int(x) for x in rng.integers(0, R, 4))):
            a, b = int(off[r]), int(off[r + 1])
            if b > a:
                rows = tab[cand[a:b]]
                if d_item == 64:
                    rows = np.concatenate([rows, np.zeros((b - a, 64), np.float32)], axis=1)
                want = o.dnn3_forward(w, prec & 1, users[r], rows) if d_item == 128 else None
                if want is None:
                    w2 = o.Dnn3Weights(d_user=128, d_item=128, h1=h1, h2=h2)
                    w2.w1 = np.concatenate([w.w1, np.zeros((64, h1), np.float32)], axis=0)
                    w2.b1, w2.w2, w2.b2, w2.w3, w2.b3 = w.b1, w.w2, w.b2, w.w3, w.b3
                    want = o.dnn3_forward(w2, prec & 1, users[r], rows)
                err = max(err, float(np.max(np.abs(got[a:b].astype(np.float64) - want.astype(np.float64)))))
        worst[prec] = max(worst[prec], err)
        cases += 1
        big = kind == 2 or uscale > 1.0                        # rows or users beyond unit norm
        # fp32: the specification itself.  bf16x3: north_star's tolerance against the FP32 oracle, nothing mirrored (1e-5; a case
        # beyond the 2e-6 regression bar is printed without failing).  Plain bf16 is only held to a looser bar against an oracle that
        # rounds where it rounds — how often it leaves 1e-5 even there is COUNTED and reported at the end.
        tol = 2e-7 if prec == 0 else ((1e-4 if big else 1.5e-5) if prec == 1 else 1e-5)
        if prec == 1:
            bf16_cases += 1
            bf16_over += err > 1e-5
        if prec == 2 and err > 2e-6:
            print("NOTE bf16x3 beyond its 2e-6 regression bar (inside north_star's 1e-5)", desc, "max abs err", err, flush=True)
        if not (err <= tol) or not np.all(np.isfinite(got)):
            bad += 1
            print("MISMATCH", desc, "max abs err", err, flush=True)
        m.destroy()
    # segmented sort
    S = int(rng.choice([1, 5, 300]))
    lens = rng.choice([0, 1, 2, 100, 5000, 16384], S)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    vals = rng.standard_normal(int(off[-1]))
    if vals.size:
        vals[rng.random(vals.size) < 0.05] = np.nan
        vals[rng.random(vals.size) < 0.02] = np.inf
        vals[rng.random(vals.size) < 0.02] = -np.inf
        vals[rng.random(vals.size) < 0.1] = 0.5
        vals[rng.random(vals.size) < 0.02] = -0.0
        desc_ = bool(rng.integers(0, 2))
        got = ctx.sort_scores(vals, off, descending=desc_)
        ok = True
        for s in sorted(set(int(x) for x in rng.integers(0, S, 6))):
            a, b = int(off[s]), int(off[s + 1])
            ok = ok and np.array_equal(got[a:b], o.sort_scores(vals[a:b], desc_))
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH sort", dict(S=S, total=int(off[-1]), descending=desc_), flush=True)
    t.destroy()
print(f"soak_rank: {cases} cases, {bad} bad; worst |error|: fp32 mode {worst[0]:.2e}, bf16 mode {worst[1]:.2e}, bf16x3 mode {worst[2]:.2e} "
      f"(bf16x3 fails at north_star's 1e-5 against the fp32 oracle); plain bf16 exceeded 1e-5 against its MIRRORING oracle in "
      f"{bf16_over} of {bf16_cases} cases", flush=True)
sys.exit(1 if bad else 0)
