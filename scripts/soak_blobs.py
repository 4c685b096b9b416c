"""Soak: pg_model_load on hostile blobs — valid DNN3 / FM + two-tower blobs truncated, extended, with header words replaced by
random values (huge / zero / negative dimensions), and pure noise: every call must come back with an error code or a working
model, never a crash.  A model that loads must rank finite scores.
Usage: soak_blobs.py [count] [seed]"""
import os, sys
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pa.Context(0)
t = pa.Table(ctx, 5000, 128)
t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights(h1=128, h2=128)
good_d = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
fw = o.Fm2tWeights(vocab=300)
good_f = pa.pack_fm2t(fw)
loaded = refused = 0
users = o.synth_rows(o.SEED_QUERY, 0, 2, 128)
cand = np.arange(200, dtype=np.uint32)
for i in range(count):
    kind = pa.MODEL_DNN3 if rng.random() < 0.5 else pa.MODEL_FM_TWOTOWER
    base = bytearray(good_d if kind == pa.MODEL_DNN3 else good_f)
    m_ = rng.integers(0, 6)
    if m_ == 0:
        base = base[: int(rng.integers(0, len(base)))]
    elif m_ == 1:
        base += bytes(rng.integers(0, 256, int(rng.integers(1, 4096)), dtype=np.uint8))
    elif m_ == 2:                                            # a header word replaced
        off = 4 * int(rng.integers(0, 16))
        val = int(rng.choice([0, 1, 64, 127, 128, 129, 1 << 20, 0x7FFFFFFF, 0xFFFFFFFF, int(rng.integers(0, 1 << 32))]))
        base[off:off + 4] = int(val).to_bytes(4, "little")
    elif m_ == 3:
        base = bytearray(bytes(rng.integers(0, 256, int(rng.integers(0, 70000)), dtype=np.uint8)))
    elif m_ == 4:                                            # the other kind's blob
        base = bytearray(good_f if kind == pa.MODEL_DNN3 else good_d)
    prec = pa.PREC_BF16 if rng.random() < 0.5 else pa.PREC_F32
    try:
        m = pa.RankModel(ctx, kind, prec, bytes(base))
    except pa._lib.PgError:
        refused += 1
        continue
    loaded += 1
    if kind == pa.MODEL_DNN3:
        try:
            s = m.rank_dnn3(t, users, cand, [0, 100, 200])
            assert np.all(np.isfinite(s)) or m_ != 5
        except pa._lib.PgError:
            pass
    m.destroy()
print(f"soak_blobs: {count} blobs, {loaded} loaded, {refused} refused, no crash", flush=True)
