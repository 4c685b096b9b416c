#!/usr/bin/env python3
"""pack_model.py — exported rank-model weights (.npz) → the blob pg_model_load takes (include/pairec_gpu.h, "rank: model
predict").  Stand-alone: numpy only, runs wherever the model was trained.

What it replaces on the reference's side: pairec never sees weights — RankConf names an algorithm whose AlgoConfs entry
points at a PAI-EAS / TF-Serving endpoint (algorithm/eas/model.go:38-120: Url, Auth, Processor, ResponseFuncName …) that
holds the SavedModel.  Here the same model's Dense kernels are loaded into HBM, so the deployment step "push the model to
EAS" becomes "pack the kernels, hand the blob to pg_model_load / ph_engine_load_dnn3[_multi]".

Input: an .npz whose arrays are the layers' kernels exactly as a framework saves them — row-major [in][out] fp32 (Keras
`dense.kernel`, PyTorch `linear.weight.T`) — and their biases:

  DNN3 (score = sigmoid(w3 . relu(W2' relu(W1' [user || item_row] + b1) + b2) + b3)):
      w1 [d_user + d_item][h1]   b1 [h1]   w2 [h1][h2]   b2 [h2]   w3 [h2] or [h2][1]   b3 scalar or [1]
  DNN3, several outputs on one trunk (probs_ctr / probs_cvr …: EasyrecResponse.multiValModule, easyrec_response.go:35-70):
      … w3 [h2][n_out]   b3 [n_out]        (n_out 2..8; the algorithm's "Outputs" list names them in this order)
  --d-user N says where the user half of w1 ends (or an integer array `d_user` inside the file); d_item = the table's dim
  (64 or 128).  --map w1=dense/kernel,b1=dense/bias,… renames arrays.  Supported hidden shapes: see the header.

  python tools/pack_model.py model.npz --d-user 128 --out model.blob      → prints kind, shape, bytes
"""
import argparse
import struct
import sys

import numpy as np

DNN3_SHAPES = {(128, 128), (256, 128), (256, 256), (512, 256), (1024, 512)}
KIND_DNN3, KIND_DNN3_MULTI = 1, 3


def _f32(a, shape=None, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (what, tuple(a.shape), tuple(shape)))
    if not np.all(np.isfinite(a)):
        raise ValueError("%s holds non-finite values" % what)
    return a


def pack_dnn3_arrays(w1, b1, w2, b2, w3, b3, d_user):
    """→ (kind, blob): PG_MODEL_DNN3 for one output, PG_MODEL_DNN3_MULTI for several (w3 [h2][n_out])."""
    w1 = np.asarray(w1)
    w2 = np.asarray(w2)
    if w1.ndim != 2 or w2.ndim != 2:
        raise ValueError("w1 / w2 must be 2-D [in][out] kernels")
    din, h1 = w1.shape
    if w2.shape[0] != h1:
        raise ValueError("w2 has shape %s: its input width must be w1's output width %d (a [out][in] kernel, e.g. PyTorch's "
                         "linear.weight, has to be transposed)" % (tuple(w2.shape), h1))
    h2 = w2.shape[1]
    d_user = int(d_user)
    d_item = din - d_user
    if not 1 <= d_user <= 4096 or d_item not in (64, 128):
        raise ValueError("w1 has %d input rows: with d_user = %d the item half is %d wide (the table's dim: 64 or 128)" % (din, d_user, d_item))
    if (h1, h2) not in DNN3_SHAPES:
        raise ValueError("hidden widths %d-%d have no kernel (supported: %s)" % (h1, h2, sorted(DNN3_SHAPES)))
    w1, w2 = _f32(w1, what="w1"), _f32(w2, (h1, h2), "w2")
    b1, b2 = _f32(np.reshape(b1, -1), (h1,), "b1"), _f32(np.reshape(b2, -1), (h2,), "b2")
    w3 = np.asarray(w3, dtype=np.float32)
    if w3.ndim == 1:
        w3 = w3.reshape(-1, 1)
    if w3.ndim != 2 or w3.shape[0] != h2:
        raise ValueError("w3 has shape %s, expected [%d] or [%d][n_out]" % (tuple(np.shape(w3)), h2, h2))
    n_out = w3.shape[1]
    b3 = _f32(np.reshape(b3, -1), (n_out,), "b3")
    w3 = _f32(w3, what="w3")
    if n_out == 1:
        blob = (struct.pack("<4I", d_user, d_item, h1, h2) + w1.tobytes() + b1.tobytes() + w2.tobytes() + b2.tobytes() +
                w3.reshape(-1).tobytes() + b3.tobytes())
        return KIND_DNN3, blob
    if not 2 <= n_out <= 8:
        raise ValueError("%d outputs (a multi-output DNN3 has 2..8)" % n_out)
    blob = (struct.pack("<5I", d_user, d_item, h1, h2, n_out) + w1.tobytes() + b1.tobytes() + w2.tobytes() + b2.tobytes() +
            w3.tobytes() + b3.tobytes())
    return KIND_DNN3_MULTI, blob


def pack_npz(path_or_mapping, d_user=None, rename=None):
    """Load the arrays (an .npz path or a mapping name → array) and pack them; → (kind, blob, description)."""
    z = np.load(path_or_mapping) if isinstance(path_or_mapping, (str, bytes)) else path_or_mapping
    rename = rename or {}

    def get(name):
        key = rename.get(name, name)
        if key not in z:
            raise KeyError("array \"%s\" not in the file (have: %s)" % (key, ", ".join(sorted(z.keys()))))
        return z[key]
    if d_user is None:
        d_user = int(np.asarray(get("d_user")).reshape(-1)[0])
    kind, blob = pack_dnn3_arrays(get("w1"), get("b1"), get("w2"), get("b2"), get("w3"), get("b3"), d_user)
    w1, w2 = get("w1"), get("w2")
    n_out = 1 if kind == KIND_DNN3 else int(np.asarray(get("b3")).size)
    desc = "%s [%d+%d]-%d-%d-%d, %d bytes" % ("PG_MODEL_DNN3" if kind == KIND_DNN3 else "PG_MODEL_DNN3_MULTI", d_user,
                                              w1.shape[0] - d_user, w1.shape[1], w2.shape[1], n_out, len(blob))
    return kind, blob, desc


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("npz")
    ap.add_argument("--d-user", type=int, default=None)
    ap.add_argument("--map", default="", help="w1=dense/kernel,b1=dense/bias,…")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    rename = dict(kv.split("=", 1) for kv in a.map.split(",") if kv)
    try:
        kind, blob, desc = pack_npz(a.npz, a.d_user, rename)
    except (KeyError, ValueError) as e:
        sys.stderr.write("pack_model: %s\n" % e)
        return 2
    with open(a.out, "wb") as f:
        f.write(blob)
    print("%s -> %s (pg_model_kind %d)" % (desc, a.out, kind))
    return 0


if __name__ == "__main__":
    sys.exit(main())
