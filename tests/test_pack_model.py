"""tools/pack_model.py: exported [in][out] kernels (.npz) → pg_model_load blobs.  CPU: the blobs are byte-identical to the
binding's packers and malformed exports are refused with the reason; GPU: a model that did NOT come from the oracle's
generator (other user width, Gaussian kernels, a bias-free layer) loads and scores within the f32 mode's 2e-7."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pack_model  # noqa: E402

import pairec_amd as pa  # noqa: E402
from oracle import oracle as o  # noqa: E402


def _export(rng, d_user, d_item, h1, h2, n_out):
    s_ = lambda *sh: (rng.standard_normal(sh) / np.sqrt(sh[0])).astype(np.float32)   # noqa: E731
    return {"w1": s_(d_user + d_item, h1), "b1": (0.1 * rng.standard_normal(h1)).astype(np.float32), "w2": s_(h1, h2),
            "b2": np.zeros(h2, np.float32), "w3": s_(h2, n_out) if n_out > 1 else s_(h2, 1)[:, 0],
            "b3": (0.1 * rng.standard_normal(n_out)).astype(np.float32)}


def test_blobs_equal_the_bindings_packers(tmp_path):
    rng = np.random.default_rng(0)
    e = _export(rng, 200, 128, 256, 128, 1)
    kind, blob, desc = pack_model.pack_npz(e, d_user=200)
    assert kind == pa.MODEL_DNN3 and blob == pa.pack_dnn3(e["w1"], e["b1"], e["w2"], e["b2"], e["w3"], float(e["b3"][0]), 200)
    assert "[200+128]-256-128-1" in desc
    e = _export(rng, 64, 64, 128, 128, 3)
    kind, blob, _ = pack_model.pack_npz(e, d_user=64)
    assert kind == pa.MODEL_DNN3_MULTI and blob == pa.pack_dnn3_multi(e["w1"], e["b1"], e["w2"], e["b2"], e["w3"], e["b3"], 64)
    # through the command line, with renamed arrays and d_user inside the file
    p = str(tmp_path / "m.npz")
    np.savez(p, **{"dense/kernel": e["w1"], "b1": e["b1"], "w2": e["w2"], "b2": e["b2"], "w3": e["w3"], "b3": e["b3"], "d_user": np.array([64])})
    out = str(tmp_path / "m.blob")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_model.py"), p, "--map", "w1=dense/kernel", "--out", out],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "PG_MODEL_DNN3_MULTI [64+64]-128-128-3" in r.stdout and open(out, "rb").read() == blob


def test_malformed_exports_are_refused(tmp_path):
    rng = np.random.default_rng(1)
    e = _export(rng, 128, 128, 512, 256, 1)
    for mutate, msg in ((lambda d: d.update(w2=d["w2"].T.copy()), "w2 has shape"),          # a [out][in] kernel (PyTorch's layout)
                        (lambda d: d.update(b1=d["b1"][:-1]), "b1 has shape"),
                        (lambda d: d.update(w1=d["w1"][:, :300], b1=d["b1"][:300], w2=d["w2"][:300]), "have no kernel"),
                        (lambda d: d["w1"].__setitem__((0, 0), np.nan), "non-finite"),
                        (lambda d: d.update(w3=np.zeros((256, 9), np.float32), b3=np.zeros(9, np.float32)), "9 outputs")):
        d = {k: v.copy() for k, v in e.items()}
        mutate(d)
        with pytest.raises(ValueError) as ei:
            pack_model.pack_npz(d, d_user=128)
        assert msg in str(ei.value)
    with pytest.raises(ValueError) as ei:
        pack_model.pack_npz(e, d_user=100)                   # the item half would be 156 wide
    assert "the table's dim" in str(ei.value)
    p = str(tmp_path / "x.npz")
    np.savez(p, w1=e["w1"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_model.py"), p, "--d-user", "128", "--out", str(tmp_path / "x.blob")],
                       capture_output=True, text=True)
    assert r.returncode == 2 and 'array "b1" not in the file' in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n_out", [1, 2])
def test_exported_model_loads_and_scores(ctx, n_out):
    rng = np.random.default_rng(5 + n_out)
    d_user, n = 200, 30_000
    e = _export(rng, d_user, 128, 256, 128, n_out)
    kind, blob, _ = pack_model.pack_npz(e, d_user=d_user)
    t = pa.Table(ctx, n, 128)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, 128)
    users = rng.standard_normal((3, d_user)).astype(np.float32)
    users /= np.linalg.norm(users, axis=1, keepdims=True)
    sizes = [900, 1, 130]
    cands = [rng.integers(0, n, s_).astype(np.uint32) for s_ in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    m = pa.RankModel(ctx, kind, pa.PREC_F32, blob)
    got = np.atleast_2d(m.rank_dnn3(t, users, np.concatenate(cands), off))
    for hd in range(n_out):
        w = o.Dnn3Weights.__new__(o.Dnn3Weights)                          # the oracle's specification on the exported arrays
        w.d_user, w.d_item, w.h1, w.h2 = d_user, 128, 256, 128
        w.w1, w.b1, w.w2, w.b2 = e["w1"], e["b1"], e["w2"], e["b2"]
        w.w3 = np.ascontiguousarray(e["w3"][:, hd] if n_out > 1 else e["w3"])
        w.b3 = float(e["b3"][hd])
        ref = np.concatenate([o.dnn3_forward(w, 0, users[r], tab[cands[r]]) for r in range(3)])
        assert np.max(np.abs(got[hd].astype(np.float64) - ref)) <= 2e-7
    m.destroy()
    t.destroy()
