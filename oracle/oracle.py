"""ctypes front-end for the CPU oracle (oracle/oracle.c) + pure-Python restatements of the
host-side pieces of pairec's rank+recall path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pairec_amd/ imports this module.

Parity status: expression fusion / sort order / decoder widening are pinned by the reference's own
known-answer tests (tests/golden/reference_known_answers.json); top-K, DNN/FM predict and DPP are
**parity unpinned** (no reference-side golden data exists).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u64, u32, i32, f32p = C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_float)
        L.orc_splitmix64.restype = u64
        L.orc_splitmix64.argtypes = [u64]
        L.orc_synth_value.restype = C.c_float
        L.orc_synth_value.argtypes = [u64, u64, u32, u32]
        L.orc_synth_rows.argtypes = [u64, u64, u64, u32, i32, f32p]
        L.orc_synth_mixture_rows.argtypes = [u64, u64, u64, u32, u32, C.c_float, u64, f32p]
        L.orc_synth_mixture_rows.restype = None
        L.orc_synth_uniform.argtypes = [u64, u64, C.c_float, f32p]
        L.orc_topk_key.restype = u64
        L.orc_topk_key.argtypes = [C.c_float, u32]
        L.orc_dot_scores.argtypes = [f32p, u64, u32, f32p, u32, f32p, i32]
        L.orc_recall_topk.restype = u32
        L.orc_recall_topk.argtypes = [f32p, u64, u32, u64, f32p, u32, u32, C.POINTER(u64), f32p, i32]
        L.orc_recall_topk_l2.restype = u32
        L.orc_recall_topk_l2.argtypes = [f32p, u64, u32, u64, f32p, u32, u32, C.POINTER(u64), f32p, i32]
        L.orc_topk_merge.restype = u32
        L.orc_topk_merge.argtypes = [C.POINTER(u64), f32p, u32, u32, u32, C.POINTER(u64), f32p]
        L.orc_dnn3_forward.argtypes = [C.c_void_p, i32, f32p, f32p, u64, f32p, i32]
        L.orc_fm2t_forward.argtypes = [C.c_void_p, i32, C.c_void_p, C.c_void_p, f32p,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64, f32p, i32]
        L.orc_fm2t_user_embedding.argtypes = [C.c_void_p, i32, f32p, f32p]
        L.orc_f32_to_bf16.restype = C.c_uint16
        L.orc_f32_to_bf16.argtypes = [C.c_float]
        L.orc_bf16_to_f32.restype = C.c_float
        L.orc_bf16_to_f32.argtypes = [C.c_uint16]
        L.orc_sort_scores.argtypes = [C.POINTER(C.c_double), u32, i32, C.POINTER(u32)]
        L.orc_dpp_kernel_matrix.argtypes = [C.POINTER(C.c_double), u32, u32, C.POINTER(C.c_double),
                                            C.c_double, C.POINTER(C.c_double)]
        L.orc_dpp_kernel_matrix_f.argtypes = [C.POINTER(C.c_double), u32, u32, C.POINTER(C.c_double), C.c_double,
                                              C.POINTER(C.c_double)]
        L.orc_dpp_with_window.restype = u32
        L.orc_dpp_with_window.argtypes = [C.POINTER(C.c_double), u32, u32, u32, C.POINTER(u32)]
        L.orc_l2_normalize_f64.argtypes = [C.POINTER(C.c_double), u32]
        L.orc_ssd_quality.restype = i32
        L.orc_ssd_quality.argtypes = [C.POINTER(C.c_double), u32, i32, C.POINTER(C.c_double)]
        L.orc_ssd_window.restype = u32
        L.orc_ssd_window.argtypes = [C.POINTER(C.c_double), u32, u32, C.POINTER(C.c_double), C.c_double,
                                     u32, u32, i32, C.POINTER(u32)]
        _lib = L
    return _lib


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------------------------------------
# synthetic data
# ---------------------------------------------------------------------------------------------
SEED_TABLE, SEED_QUERY, SEED_WEIGHTS, SEED_FIELDS, SEED_CANDS = (
    0x5EED0001, 0x5EED0002, 0x5EED0003, 0x5EED0004, 0x5EED0005)


def synth_rows(seed: int, row0: int, nrows: int, dim: int, normalize: bool = True) -> np.ndarray:
    out = np.empty((nrows, dim), dtype=np.float32)
    lib().orc_synth_rows(seed, row0, nrows, dim, int(normalize), _f32p(out))
    return out


def synth_mixture_rows(seed: int, row0: int, nrows: int, dim: int, n_centres: int, sigma: float, stream: int = 0) -> np.ndarray:
    """Clustered rows (pg_table_fill_mixture's definition, bit for bit at stream 0): n_centres centres on the unit sphere,
    within-cluster noise of norm ~ sigma, normalised; stream > 0 draws further points of the same mixture (queries)."""
    out = np.empty((nrows, dim), dtype=np.float32)
    ns = np.float32(sigma) * np.sqrt(np.float32(3.0) / np.float32(dim), dtype=np.float32)
    lib().orc_synth_mixture_rows(seed, row0, nrows, dim, n_centres, float(np.float32(ns)), stream, _f32p(out))
    return out


def synth_uniform(seed: int, n: int, scale: float = 1.0) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().orc_synth_uniform(seed, n, scale, _f32p(out))
    return out


def splitmix64(x: int) -> int:
    return int(lib().orc_splitmix64(x & 0xFFFFFFFFFFFFFFFF))


# ---------------------------------------------------------------------------------------------
# recall
# ---------------------------------------------------------------------------------------------
def dot_scores(table: np.ndarray, queries: np.ndarray, threads: int = 0) -> np.ndarray:
    table = np.ascontiguousarray(table, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, table.shape[1])
    out = np.empty((queries.shape[0], table.shape[0]), dtype=np.float32)
    lib().orc_dot_scores(_f32p(table), table.shape[0], table.shape[1], _f32p(queries),
                         queries.shape[0], _f32p(out), threads)
    return out


def recall_topk(table: np.ndarray, queries: np.ndarray, k: int, row_offset: int = 0,
                threads: int = 0):
    table = np.ascontiguousarray(table, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, table.shape[1])
    nq = queries.shape[0]
    rows = np.zeros((nq, k), dtype=np.uint64)
    scores = np.zeros((nq, k), dtype=np.float32)
    n = lib().orc_recall_topk(_f32p(table), table.shape[0], table.shape[1], row_offset,
                              _f32p(queries), nq, k, rows.ctypes.data_as(C.POINTER(C.c_uint64)),
                              _f32p(scores), threads)
    return rows[:, :n], scores[:, :n]


def recall_topk_l2(table: np.ndarray, queries: np.ndarray, k: int, row_offset: int = 0, threads: int = 0):
    """Top-k by SMALLEST squared Euclidean distance (service/recall/hologres_vector_recall_v2.go:23: ORDER BY
    pm_approx_squared_euclidean_distance ascending; the distance is the item's score, :181-189).  Specified as
    d = fmaf(-2, ip, |x|^2 + |q|^2), every sum a k-ascending fp32 fmaf chain; ties by row ascending."""
    table = np.ascontiguousarray(table, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, table.shape[1])
    nq = queries.shape[0]
    rows = np.zeros((nq, k), dtype=np.uint64)
    dist = np.zeros((nq, k), dtype=np.float32)
    n = lib().orc_recall_topk_l2(_f32p(table), table.shape[0], table.shape[1], row_offset,
                                 _f32p(queries), nq, k, rows.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 _f32p(dist), threads)
    return rows[:, :n], dist[:, :n]


def topk_merge(rows: np.ndarray, scores: np.ndarray, k: int):
    """rows/scores: [nlists, per_list] → global top-k (score desc, row asc)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    orow = np.zeros(k, dtype=np.uint64)
    osc = np.zeros(k, dtype=np.float32)
    n = lib().orc_topk_merge(rows.ctypes.data_as(C.POINTER(C.c_uint64)), _f32p(scores),
                             rows.shape[0], rows.shape[1], k,
                             orow.ctypes.data_as(C.POINTER(C.c_uint64)), _f32p(osc))
    return orow[:n], osc[:n]


def topk_key(score: float, row: int) -> int:
    return int(lib().orc_topk_key(score, row))


# ---------------------------------------------------------------------------------------------
# rank models
# ---------------------------------------------------------------------------------------------
class _Dnn3(C.Structure):
    _fields_ = [("d_user", C.c_uint32), ("d_item", C.c_uint32), ("h1", C.c_uint32),
                ("h2", C.c_uint32), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_float)]


class Dnn3Weights:
    """3-layer MLP [user‖item] -> h1 (ReLU) -> h2 (ReLU) -> 1 (sigmoid).  cfg 3: 256->512->256->1."""

    def __init__(self, d_user=128, d_item=128, h1=512, h2=256, seed=SEED_WEIGHTS):
        self.d_user, self.d_item, self.h1, self.h2 = d_user, d_item, h1, h2
        din = d_user + d_item
        # uniform ±1/sqrt(fan_in); independent streams per tensor
        self.w1 = synth_uniform(seed ^ 0x101, din * h1, 1.0 / math.sqrt(din)).reshape(din, h1)
        self.b1 = synth_uniform(seed ^ 0x102, h1, 1.0 / math.sqrt(din))
        self.w2 = synth_uniform(seed ^ 0x103, h1 * h2, 1.0 / math.sqrt(h1)).reshape(h1, h2)
        self.b2 = synth_uniform(seed ^ 0x104, h2, 1.0 / math.sqrt(h1))
        self.w3 = synth_uniform(seed ^ 0x105, h2, 1.0 / math.sqrt(h2))
        self.b3 = float(synth_uniform(seed ^ 0x106, 1, 1.0 / math.sqrt(h2))[0])

    def _struct(self):
        s = _Dnn3(self.d_user, self.d_item, self.h1, self.h2, self.w1.ctypes.data,
                  self.b1.ctypes.data, self.w2.ctypes.data, self.b2.ctypes.data,
                  self.w3.ctypes.data, self.b3)
        return s


def dnn3_forward(w: Dnn3Weights, prec: int, user_vec: np.ndarray, item_rows: np.ndarray,
                 threads: int = 0) -> np.ndarray:
    item_rows = np.ascontiguousarray(item_rows, dtype=np.float32)
    user_vec = np.ascontiguousarray(user_vec, dtype=np.float32)
    out = np.empty(item_rows.shape[0], dtype=np.float32)
    s = w._struct()
    lib().orc_dnn3_forward(C.byref(s), prec, _f32p(user_vec), _f32p(item_rows),
                           item_rows.shape[0], _f32p(out), threads)
    return out


class Dnn3MultiWeights(Dnn3Weights):
    """A multi-output model: n_out heads (probs_ctr, probs_cvr, ...) on ONE shared trunk — the shape of the reference's
    own fixtures (algorithm/eas/easyrec_response.go:35-70, utils/ast/ast_test.go:90-129: ppnet_probs_ctr / _cvr).
    w3 [h2][n_out], b3 [n_out]; head o is the DNN3 whose last layer is (w3[:, o], b3[o])."""

    def __init__(self, n_out=2, d_user=128, d_item=128, h1=512, h2=256, seed=SEED_WEIGHTS):
        super().__init__(d_user, d_item, h1, h2, seed)
        self.n_out = n_out
        cols = [self.w3] + [synth_uniform(seed ^ (0x305 + 16 * o), h2, 1.0 / math.sqrt(h2)) for o in range(1, n_out)]
        self.w3m = np.ascontiguousarray(np.stack(cols, axis=1))                       # [h2][n_out]
        self.b3m = np.array([self.b3] + [float(synth_uniform(seed ^ (0x306 + 16 * o), 1, 1.0 / math.sqrt(h2))[0])
                                         for o in range(1, n_out)], dtype=np.float32)

    def head(self, o: int) -> Dnn3Weights:
        h = Dnn3Weights.__new__(Dnn3Weights)
        h.__dict__.update({k_: v for k_, v in self.__dict__.items() if k_ not in ("w3m", "b3m", "n_out")})
        h.w3 = np.ascontiguousarray(self.w3m[:, o])
        h.b3 = float(self.b3m[o])
        return h


def dnn3_multi_forward(w: Dnn3MultiWeights, prec: int, user_vec: np.ndarray, item_rows: np.ndarray, threads: int = 0) -> np.ndarray:
    """scores [n_out][n_items]: every head through the single-output specification (orc_dnn3_forward) with its own column."""
    return np.stack([dnn3_forward(w.head(o), prec, user_vec, item_rows, threads) for o in range(w.n_out)])


class _Fm2t(C.Structure):
    _fields_ = [("n_user_fields", C.c_uint32), ("n_item_fields", C.c_uint32), ("k", C.c_uint32),
                ("d_user", C.c_uint32), ("t_h1", C.c_uint32), ("t_out", C.c_uint32),
                ("fm_w", C.c_void_p), ("fm_b", C.c_float),
                ("uw1", C.c_void_p), ("ub1", C.c_void_p), ("uw2", C.c_void_p), ("ub2", C.c_void_p),
                ("iw1", C.c_void_p), ("ib1", C.c_void_p), ("iw2", C.c_void_p), ("ib2", C.c_void_p)]


class Fm2tWeights:
    """FM (16 fields × k=16) + two-tower (128->256->64 each side), SURVEY.md §8(d) cfg 4."""

    def __init__(self, n_user_fields=8, n_item_fields=8, k=16, d_user=128, t_h1=256, t_out=64,
                 vocab=1_000_000, seed=SEED_WEIGHTS, field_seed=SEED_FIELDS):
        self.nuf, self.nif, self.k = n_user_fields, n_item_fields, k
        self.d_user, self.t_h1, self.t_out, self.vocab = d_user, t_h1, t_out, vocab
        din = n_item_fields * k
        self.fm_b = float(synth_uniform(seed ^ 0x200, 1, 0.1)[0])
        self.uw1 = synth_uniform(seed ^ 0x201, d_user * t_h1, 1 / math.sqrt(d_user)).reshape(d_user, t_h1)
        self.ub1 = synth_uniform(seed ^ 0x202, t_h1, 1 / math.sqrt(d_user))
        self.uw2 = synth_uniform(seed ^ 0x203, t_h1 * t_out, 1 / math.sqrt(t_h1)).reshape(t_h1, t_out)
        self.ub2 = synth_uniform(seed ^ 0x204, t_out, 1 / math.sqrt(t_h1))
        self.iw1 = synth_uniform(seed ^ 0x205, din * t_h1, 1 / math.sqrt(din)).reshape(din, t_h1)
        self.ib1 = synth_uniform(seed ^ 0x206, t_h1, 1 / math.sqrt(din))
        self.iw2 = synth_uniform(seed ^ 0x207, t_h1 * t_out, 1 / math.sqrt(t_h1)).reshape(t_h1, t_out)
        self.ib2 = synth_uniform(seed ^ 0x208, t_out, 1 / math.sqrt(t_h1))
        nf = n_user_fields + n_item_fields
        # field embedding tables: unnormalised uniform ±0.25; linear tables ±0.1
        self.field_emb = [synth_rows(field_seed + 16 * f, 0, vocab, k, normalize=False) * np.float32(0.25)
                          for f in range(nf)]
        self.field_lin = [synth_uniform((field_seed + 16 * f) ^ 0xABCD, vocab, 0.1) for f in range(nf)]

    def _struct(self):
        return _Fm2t(self.nuf, self.nif, self.k, self.d_user, self.t_h1, self.t_out, None, self.fm_b,
                     self.uw1.ctypes.data, self.ub1.ctypes.data, self.uw2.ctypes.data,
                     self.ub2.ctypes.data, self.iw1.ctypes.data, self.ib1.ctypes.data,
                     self.iw2.ctypes.data, self.ib2.ctypes.data)


def fm2t_forward(w: Fm2tWeights, prec: int, user_vec, user_field_ids, item_field_ids,
                 threads: int = 0) -> np.ndarray:
    user_vec = np.ascontiguousarray(user_vec, dtype=np.float32)
    uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
    itf = np.ascontiguousarray(item_field_ids, dtype=np.int32).reshape(-1, w.nif)
    nf = w.nuf + w.nif
    emb_ptrs = (C.c_void_p * nf)(*[a.ctypes.data for a in w.field_emb])
    lin_ptrs = (C.c_void_p * nf)(*[a.ctypes.data for a in w.field_lin])
    out = np.empty(itf.shape[0], dtype=np.float32)
    s = w._struct()
    lib().orc_fm2t_forward(C.byref(s), prec, emb_ptrs, lin_ptrs, _f32p(user_vec),
                           uf.ctypes.data_as(C.POINTER(C.c_int32)),
                           itf.ctypes.data_as(C.POINTER(C.c_int32)), itf.shape[0], _f32p(out),
                           threads)
    return out


def fm2t_user_embedding(w: Fm2tWeights, prec: int, user_vec) -> np.ndarray:
    """User-tower output [t_out] — the user embedding of OnlineVectorRecall (online_vector_recall.go:97-109)."""
    user_vec = np.ascontiguousarray(user_vec, dtype=np.float32)
    out = np.empty(w.t_out, dtype=np.float32)
    s = w._struct()
    lib().orc_fm2t_user_embedding(C.byref(s), prec, _f32p(user_vec), _f32p(out))
    return out


def f32_to_bf16_round(a: np.ndarray) -> np.ndarray:
    """RNE bf16 rounding, returned as fp32 (vectorised twin of orc_f32_to_bf16)."""
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (b & 0x7FFFFFFF) > 0x7F800000
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    r = np.where(nan, ((b >> 16) | 0x40) << 16, r)
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(np.shape(a))


# ---------------------------------------------------------------------------------------------
# sorts  (sort/item_score.go:15-18, :36-41; sort/item_rank_score.go:26-32)
# ---------------------------------------------------------------------------------------------
def sort_scores(scores: Sequence[float], desc: bool) -> np.ndarray:
    s = np.ascontiguousarray(scores, dtype=np.float64)
    out = np.empty(s.shape[0], dtype=np.uint32)
    if s.shape[0]:
        lib().orc_sort_scores(_f64p(s), s.shape[0], int(desc), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


# ---------------------------------------------------------------------------------------------
# DPP  (sort/dpp_sort.go:372-551)
# ---------------------------------------------------------------------------------------------
def l2_normalize_f64(v: np.ndarray) -> np.ndarray:
    v = np.array(v, dtype=np.float64, copy=True)
    for row in v.reshape(-1, v.shape[-1]):
        lib().orc_l2_normalize_f64(_f64p(row), row.shape[0])
    return v


def dpp_kernel_matrix(emb: np.ndarray, rel: np.ndarray, alpha: float) -> np.ndarray:
    emb = np.ascontiguousarray(emb, dtype=np.float64)
    rel = np.ascontiguousarray(rel, dtype=np.float64)
    n, d = emb.shape
    L = np.empty((n, n), dtype=np.float64)
    lib().orc_dpp_kernel_matrix(_f64p(emb), n, d, _f64p(rel), alpha, _f64p(L))
    return L


def dpp_relevance(rel: np.ndarray, mode: int):
    """dpp_norm_relevance_score (dpp_sort.go:382-405): the same two normalisations SSDSort applies to its quality
    scores (z-score by stat.PopMeanVariance / StdScore; min-max with max = first, min = last item).
    Returns (scores, ok); ok False = "all item score is zero" (the reference returns the items unchanged)."""
    return ssd_quality(rel, mode)


def dpp_features(emb32, hook, normalize: bool, ensure_pos_similarity: bool) -> np.ndarray:
    """KernelMatrix's feature rows (dpp_sort.go:408-447).  emb32 [n][d] fp32 item embeddings (the table path;
    None = hook embeddings only), hook [n][h] fp64 (RegisterEmbeddingHook outputs; None = none).
    Table path: e = emb (L2-normalised when `normalize`, loadEmbeddingCache :235-236); with hooks the row is
    [hook ‖ e] re-normalised jointly (:419-421); then append 1 and scale by 1/sqrt 2 (:428-430, EnsurePositiveSim
    is not consulted).  Hook-only path (:434-447): normalise when asked; ensure_pos → [c,1]/sqrt 2, else [c,0]."""
    isq2 = 0.70710678118654757
    has_table = emb32 is not None
    n = (emb32 if has_table else hook).shape[0]
    rows = []
    for i in range(n):
        c = [] if hook is None else [float(x) for x in np.asarray(hook[i], dtype=np.float64)]
        if has_table:
            e = np.asarray(emb32[i], dtype=np.float32).astype(np.float64)
            if normalize:
                e = l2_normalize_f64(e[None])[0]
            c = c + [float(x) for x in e]
            renorm = hook is not None
        else:
            renorm = normalize
        c = np.array(c, dtype=np.float64)
        if renorm:
            c = l2_normalize_f64(c[None])[0]
        if has_table or ensure_pos_similarity:
            c = np.concatenate([c * isq2, [isq2 * 1.0]])
        else:
            c = np.concatenate([c, [0.0]])
        rows.append(c)
    return np.ascontiguousarray(np.stack(rows))


def dpp_kernel_matrix_f(F: np.ndarray, rel: np.ndarray, alpha: float) -> np.ndarray:
    F = np.ascontiguousarray(F, dtype=np.float64)
    rel = np.ascontiguousarray(rel, dtype=np.float64)
    n, d1 = F.shape
    L = np.empty((n, n), dtype=np.float64)
    lib().orc_dpp_kernel_matrix_f(_f64p(F), n, d1, _f64p(rel), alpha, _f64p(L))
    return L


def dpp_with_window(L: np.ndarray, topn: int, window: int) -> np.ndarray:
    L = np.ascontiguousarray(L, dtype=np.float64)
    out = np.zeros(max(topn, 1), dtype=np.uint32)
    n = lib().orc_dpp_with_window(_f64p(L), L.shape[0], topn, window,
                                  out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out[:n]


# ---------------------------------------------------------------------------------------------
# SSD  (sort/ssd_sort.go:346-486)
# ---------------------------------------------------------------------------------------------
def ssd_quality(rel: np.ndarray, mode: int):
    """ssd_norm_quality_score (:360-388).  Returns (scores, ok); ok False = the reference bails out."""
    rel = np.ascontiguousarray(rel, dtype=np.float64)
    out = np.empty_like(rel)
    ok = lib().orc_ssd_quality(_f64p(rel), rel.shape[0], int(mode), _f64p(out))
    return out, bool(ok)


def ssd_embeddings(emb32: np.ndarray, normalize: bool, ensure_pos_similarity: bool) -> np.ndarray:
    """loadEmbeddingCache's per-item treatment (:246-252): widen, L2-normalise, append 1."""
    e = np.array(emb32, dtype=np.float64)
    if normalize:
        e = l2_normalize_f64(e)
    if ensure_pos_similarity:
        e = np.concatenate([e, np.ones((e.shape[0], 1))], axis=1)
    return np.ascontiguousarray(e)


def ssd_window(emb: np.ndarray, rel: np.ndarray, gamma: float, topn: int, window: int,
               use_ssd_star: bool = False) -> np.ndarray:
    """SSDWithSlidingWindow on prepared fp64 embeddings [n][d] and (normalised) quality scores."""
    e = np.array(emb, dtype=np.float64, copy=True, order="C")
    rel = np.ascontiguousarray(rel, dtype=np.float64)
    out = np.zeros(max(min(topn, e.shape[0]), 1), dtype=np.uint32)
    n = lib().orc_ssd_window(_f64p(e), e.shape[0], e.shape[1], _f64p(rel), float(gamma), int(topn), int(window),
                             int(use_ssd_star), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out[:n]


# ---------------------------------------------------------------------------------------------
# score-fusion expression language  (utils/ast/parse.go:57-158, utils/ast/ast.go:81-268)
# ---------------------------------------------------------------------------------------------
LITERAL, OPERATOR, PARAMETER = 0, 1, 2
_PRECEDENCE = {"+": 20, "-": 20, "*": 40, "/": 40, "%": 40, "^": 60, "#": 80}   # ast.go:81


class ExprError(Exception):
    pass


def expr_tokenize(src: str):
    """Lexer restating parse.go:40-158: operators `# ( ) + - * / ^ %`; literals start with a digit
    and extend over [0-9._e] (underscores stripped); `${name}` parameters; any other character is
    a "symbol error".  Quirks kept: a `$` not followed by `{` silently ends lexing; trailing
    whitespace is fine only if the last whitespace character is a plain space (the reference
    re-examines its stale `ch` after running off the end, parse.go:66-72,125-133)."""
    if src == "":
        raise ExprError("empty source")      # the reference indexes s[0] (parse.go:176)
    toks = []
    n = len(src)
    off = 0
    ws = " \t\n\v\f\r"
    while off < n:
        last_ws = None
        while off < n and src[off] in ws:
            last_ws = src[off]
            off += 1
        if off >= n:
            if last_ws is not None and last_ws != " ":
                raise ExprError("symbol error: unknown %r" % (last_ws,))
            break
        ch = src[off]
        start = off
        if ch in "#()+-*/^%":
            toks.append((ch, OPERATOR, start))
            off += 1
        elif "0" <= ch <= "9":
            while off < n and ("0" <= src[off] <= "9" or src[off] in "._e"):
                off += 1
            toks.append((src[start:off].replace("_", ""), LITERAL, start))
        elif ch == "$":
            off += 1
            if off < n and src[off] == "{":
                while off < n and src[off] != "}":
                    off += 1
                toks.append((src[start + 2:off], PARAMETER, start + 2))
                off += 1
            else:
                break
        else:
            raise ExprError("symbol error: unknown %r, pos [%d:]" % (ch, start))
    return toks


def _go_parse_float(tok: str) -> Optional[float]:
    """strconv.ParseFloat(tok, 64) over the literal alphabet [0-9.e]; None where Go returns an
    error (syntax error, or ErrRange for overflow to ±Inf)."""
    import re
    if not re.fullmatch(r"(\d+\.?\d*)(e\d+)?", tok):
        return None
    v = float(tok)
    if math.isinf(v):
        return None
    return v


class _Parser:
    """Precedence-climbing parser restating ast.go:84-197, including its error quirks: a malformed
    literal (or an operator in primary position, e.g. unary minus) yields Number(0) *without
    consuming the token* and records Err, which GetExpAST ignores (ast.go:368-389) — so "-5"
    evaluates as 0-5 and "2*1e-5" as 2*0.  After the last token is consumed `currTok` stays on it
    (getNextToken returns nil without moving currTok, ast.go:90-97)."""

    def __init__(self, toks):
        if not toks:
            raise ExprError("empty token")
        self.toks = toks
        self.i = 0
        self.err = None

    def cur(self):
        return self.toks[min(self.i, len(self.toks) - 1)]

    def next(self):
        self.i = min(self.i + 1, len(self.toks))
        return self.toks[self.i] if self.i < len(self.toks) else None

    def prec(self):
        return _PRECEDENCE.get(self.cur()[0], -1)    # keyed on the token text only (ast.go:100-105)

    def parse_number(self):
        t = self.cur()
        v = _go_parse_float(t[0])
        if v is None:
            self.err = "want '(' or '0-9' but get %r" % (t[0],)
            return ("num", 0.0)
        self.next()
        return ("num", v)

    def parse_primary(self):
        t = self.cur()
        if t[1] == LITERAL:
            return self.parse_number()
        if t[1] == PARAMETER:
            self.next()
            return ("param", t[0])
        if t[0] == "(":
            self.next()
            e = self.parse_expression()
            if e is None:
                return None
            if self.cur()[0] != ")":
                self.err = "want ')' but get %s" % (self.cur()[0],)
                return None
            self.next()
            return e
        return self.parse_number()

    def parse_expression(self):
        return self.parse_binop_rhs(0, self.parse_primary())

    def parse_binop_rhs(self, exec_prec, lhs):
        while True:
            tp = self.prec()
            if tp < exec_prec:
                return lhs
            op = self.cur()[0]
            if self.next() is None:
                return lhs
            rhs = self.parse_primary()
            if rhs is None:
                return None
            if tp < self.prec():
                rhs = self.parse_binop_rhs(tp + 1, rhs)
                if rhs is None:
                    return None
            lhs = ("bin", op, lhs, rhs)


def expr_parse(src: str):
    """GetExpAST (ast.go:368-389): '' → None; otherwise the (possibly truncated) AST."""
    if src == "":
        return None
    return _Parser(expr_tokenize(src)).parse_expression()


def _go_int(x: float) -> int:
    # Go float64→int truncates toward zero; out-of-range/NaN is implementation-defined
    # (amd64 CVTTSD2SQ yields MinInt64)
    if x != x or abs(x) >= 2.0 ** 63:
        return -(2 ** 63)
    return int(x)


def expr_eval(ast, lookup) -> float:
    """ExprASTResult (ast.go:215-268).  lookup(name) → float or None (not found → 0.0).
    Raises ExprError on division by zero / integer modulo by zero (the reference panics)."""
    if ast is None:
        return 0.0
    kind = ast[0]
    if kind == "num":
        return ast[1]
    if kind == "param":
        v = lookup(ast[1])
        return 0.0 if v is None else float(v)
    _, op, lhs, rhs = ast
    l = expr_eval(lhs, lookup)
    r = expr_eval(rhs, lookup)
    if op == "#":
        return l if l != 0.0 else r
    if op == "^":
        return go_pow(l, r)
    if op == "+":
        return l + r
    if op == "-":
        return l - r
    if op == "*":
        return l * r
    if op == "/":
        if r == 0:
            raise ExprError("division by zero [%g/%g]" % (l, r))
        return l / r
    if op == "%":
        li, ri = _go_int(l), _go_int(r)
        if ri == 0:
            raise ExprError("integer divide by zero")
        m = abs(li) % abs(ri)                  # Go's % truncates toward zero
        return float(-m if li < 0 else m)
    return 0.0


def pow_last_ulp_explains(evaluate, device_value) -> bool:
    """The device's pow() is within 2 ulp of libm's; where a power feeds something discontinuous — an integer `%`,
    the exponent of a negative base (an integer or not: a number or NaN; even or odd: its sign) — that ulp changes the value, not its
    last digit.  True iff `evaluate()` (an oracle evaluation) reproduces `device_value` (NaN matching NaN) with the results of its
    first three go_pow calls moved by -2 … +2 ulps each, independently: the difference is the pow tolerance, not an evaluator's."""
    import itertools
    global go_pow
    orig = go_pow
    d = float(device_value)
    try:
        for shifts in itertools.product((0, -1, 1, -2, 2), repeat=3):
            if shifts == (0, 0, 0):
                continue
            calls = [0]

            def nudged(x, y):
                r = orig(x, y)
                k = shifts[calls[0]] if calls[0] < 3 else 0
                calls[0] += 1
                if k and math.isfinite(r) and r != 0.0:
                    for _ in range(abs(k)):
                        r = math.nextafter(r, math.inf if k > 0 else -math.inf)
                return r
            go_pow = nudged
            try:
                w = float(evaluate())
            except Exception:
                continue
            if (math.isnan(w) and math.isnan(d)) or w == d or (math.isfinite(w) and abs(w - d) <= 1e-10 * max(abs(w), 1e-300)):
                return True
    finally:
        go_pow = orig
    return False


def go_pow(x: float, y: float) -> float:
    """math.Pow (Go stdlib `math/pow.go`, go 1.24 per reference go.mod:3).  Go applies the INTEGER part of the exponent by
    repeated squaring of Frexp(x)'s mantissa with the binary exponent carried on the side (exact where the products are, e.g.
    400^4); integer-valued exponents follow that loop here bit for bit.  Fractional exponents (Go: Exp(yf Log(x)) times the
    integer part, with a platform-specific Exp on amd64) and the special cases are mapped onto C pow(): within 1-2 ulp of Go,
    so fused scores with a fractional `^` are compared with rel 1e-14."""
    ay = abs(y)
    if y == 1.0:
        return x
    if math.isfinite(x) and x != 0.0 and x != 1.0 and y != 0.0 and ay < 2.0 ** 63 and ay == math.floor(ay):
        a1, ae = 1.0, 0
        x1, xe = math.frexp(x)
        i = int(ay)
        while i != 0:
            if xe < -(1 << 12) or (1 << 12) < xe:
                ae += xe
                break
            if i & 1:
                a1 *= x1
                ae += xe
            x1 *= x1
            xe <<= 1
            if x1 < 0.5:
                x1 += x1
                xe -= 1
            i >>= 1
        if y < 0:
            a1 = 1.0 / a1
            ae = -ae
        ae = max(-4096, min(4096, ae))
        try:
            return math.ldexp(a1, ae)
        except OverflowError:
            return math.copysign(math.inf, a1)
    try:
        return math.pow(x, y)
    except OverflowError:
        return math.inf if (x > 0 or float(y).is_integer() and int(y) % 2 == 0) else -math.inf
    except ValueError:
        if x == 0 and y < 0:
            return math.copysign(math.inf, x) if float(y).is_integer() and int(y) % 2 == 1 else math.inf
        return math.nan


# ---------------------------------------------------------------------------------------------
# ASTType "antlr" (utils/ast/ast.go:275-389): the expression goes to go-antlr-valuate v0.0.4 — NOT vendored, its grammar is
# known here only through the reference's own tests (utils/ast/ast_test.go:30-56,90-167,213-300) and the functions pairec
# registers (utils/ast/antlr_functions.go:34-91).  Restated: the subset those pin — + - * / ^ with ^ = math.Pow binding
# tighter than * /, those tighter than + -; parentheses; unary minus; numbers; ${name}; maxIndex(${v}) / maxValue(${v}) —
# and ExprASTResultByAntlr's error rule: a variable the data map lacks (or a non-numeric one) fails Evaluate → 0.
# "parity unpinned" beyond those vectors: `/` is taken as Go's float64 division, a chained a^b^c is refused.
# ---------------------------------------------------------------------------------------------
class AntlrUnsupported(Exception):
    pass


def antlr_parse(src: str):
    """GetExpASTByAntlr for the pinned subset: '' → None; anything outside the subset raises AntlrUnsupported."""
    if src == "":
        return None
    pos = [0]

    def ws():
        while pos[0] < len(src) and src[pos[0]] in " \t\n\r":
            pos[0] += 1

    def param():
        if src[pos[0]:pos[0] + 2] != "${":
            raise AntlrUnsupported("expected ${name}")
        close = src.find("}", pos[0] + 2)
        if close < 0 or close == pos[0] + 2:
            raise AntlrUnsupported("unterminated ${")
        name = src[pos[0] + 2:close]
        pos[0] = close + 1
        return name

    def primary():
        ws()
        if pos[0] >= len(src):
            raise AntlrUnsupported("unexpected end")
        c = src[pos[0]]
        if c == "(":
            pos[0] += 1
            e = expr()
            ws()
            if pos[0] >= len(src) or src[pos[0]] != ")":
                raise AntlrUnsupported("missing )")
            pos[0] += 1
            return e
        if c == "-":
            # (prefix minus against ^: the reference's tests never combine them and govaluate-style grammars bind the prefix
            #  tighter than the exponent (-2^2 = 4) — refused rather than evaluated on an assumption; write -(a^b) or (-a)^b)
            pos[0] += 1
            operand = primary()
            ws()
            if pos[0] < len(src) and src[pos[0]] == "^":
                raise AntlrUnsupported("-a^b: parenthesise")
            return ("neg", operand)
        if c == "$":
            return ("param", param())
        if c.isdigit() or c == ".":
            m = re.match(r"(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", src[pos[0]:])
            if not m:
                raise AntlrUnsupported("malformed number")
            pos[0] += m.end()
            return ("num", float(m.group(0)))
        if c.isalpha() or c == "_":
            m = re.match(r"[A-Za-z_][A-Za-z0-9_]*", src[pos[0]:])
            fn = m.group(0)
            if fn not in ("maxIndex", "maxValue"):
                raise AntlrUnsupported("function %s" % fn)
            pos[0] += m.end()
            ws()
            if pos[0] >= len(src) or src[pos[0]] != "(":
                raise AntlrUnsupported("%s(" % fn)
            pos[0] += 1
            ws()
            name = param()
            ws()
            if pos[0] >= len(src) or src[pos[0]] != ")":
                raise AntlrUnsupported("%s: one ${list} argument" % fn)
            pos[0] += 1
            return (fn, name)
        raise AntlrUnsupported("'%s'" % c)

    def power():
        b = primary()
        ws()
        if pos[0] < len(src) and src[pos[0]] == "^":
            pos[0] += 1
            b = ("bin", "^", b, primary())
            ws()
            if pos[0] < len(src) and src[pos[0]] == "^":
                raise AntlrUnsupported("chained ^")
        return b

    def term():
        l = power()
        while True:
            ws()
            if pos[0] >= len(src) or src[pos[0]] not in "*/":
                return l
            if src[pos[0]:pos[0] + 2] == "**":
                raise AntlrUnsupported("**")
            op = src[pos[0]]
            pos[0] += 1
            l = ("bin", op, l, power())

    def expr():
        l = term()
        while True:
            ws()
            if pos[0] >= len(src) or src[pos[0]] not in "+-":
                return l
            op = src[pos[0]]
            pos[0] += 1
            l = ("bin", op, l, term())

    e = expr()
    ws()
    if pos[0] != len(src):
        raise AntlrUnsupported("'%s'" % src[pos[0]])
    return e


def antlr_result(ast, data: Dict[str, object]) -> float:
    """ExprASTResultByAntlr (ast.go:374-389) over the item's ExprData map: an evaluation error → 0."""
    class _Fail(Exception):
        pass

    def num(v):
        if isinstance(v, bool) or not isinstance(v, (int, float, np.integer, np.floating)):
            raise _Fail()
        return float(v)

    def ev(a):
        k = a[0]
        if k == "num":
            return a[1]
        if k == "param":
            if a[1] not in data:
                raise _Fail()
            return num(data[a[1]])
        if k == "neg":
            return -ev(a[1])
        if k in ("maxIndex", "maxValue"):
            if a[1] not in data or not isinstance(data[a[1]], (list, tuple, np.ndarray)) or len(data[a[1]]) == 0:
                raise _Fail()
            vals = [to_float(x, 0.0) for x in data[a[1]]]
            best = 0
            for i in range(1, len(vals)):                      # findMax (antlr_functions.go:72-91): first maximum
                if vals[i] > vals[best]:
                    best = i
            return float(best) if k == "maxIndex" else vals[best]
        l, r = ev(a[2]), ev(a[3])
        if a[1] == "+":
            return l + r
        if a[1] == "-":
            return l - r
        if a[1] == "*":
            return l * r
        if a[1] == "/":
            return float(np.float64(l) / np.float64(r)) if r != 0 else float(np.divide(np.float64(l), np.float64(r)))
        return go_pow(l, r)
    if ast is None:
        return 0.0
    try:
        with np.errstate(all="ignore"):
            return ev(ast)
    except _Fail:
        return 0.0


class OracleItem:
    """module.Item restated (module/item.go:15-27,168-212): only what the hot path touches."""

    def __init__(self, item_id: str, score: float = 0.0, retrieve_id: str = ""):
        self.id = item_id
        self.score = float(score)
        self.retrieve_id = retrieve_id
        self.item_type = ""
        self.properties: Dict[str, object] = {}
        self.algo_scores: Dict[str, float] = {}
        self.recall_scores: Optional[Dict[str, float]] = None

    def add_algo_score(self, name, score):
        self.algo_scores[name] = float(score)

    def add_property(self, name, v):
        self.properties[name] = v

    def float_expr_data(self, name):
        """Item.FloatExprData (module/item.go:189-212)."""
        if name == "current_score":
            self.algo_scores["recall_score"] = self.score
            return self.score
        if name in self.algo_scores:
            return self.algo_scores[name]
        if name in self.properties:
            return to_float(self.properties[name], 0.0)
        return None


def to_float(v, default: float) -> float:
    """utils.ToFloat (utils/type.go:43-69)."""
    if isinstance(v, bool):
        return default
    if isinstance(v, (int, float, np.integer, np.floating)):
        return float(v)
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return default
    return default


def fuse_scores(expr: str, items: List[OracleItem], ctx_params: Optional[Dict[str, float]] = None,
                score_rewrite: Optional[Dict[str, str]] = None):
    """rank_service.go:339-363: item.Score = ExprASTResult(ast, AstParameterData{ctx,item});
    AB params win when non-zero (service/rank/ast_parameter_data.go:30-40).
    score_rewrite = RankConfig.ScoreRewrite (rank_service.go:296-306,343-353): per item every source's expression is
    evaluated over the item as it stands, the results collected in a map, then written back with AddAlgoScores
    (module/item.go:177-188) before RankScore; a source whose expression does not parse scores 0 (:299-303,349-351).
    An empty RankScore skips both (:339)."""
    if expr == "":
        return
    try:
        ast = expr_parse(expr)
    except Exception:                                          # noqa: BLE001 — the reference logs and leaves exprAst nil (:291-294)
        ast = None
    ctx_params = ctx_params or {}
    rewrite_asts = {}
    for source, src_expr in (score_rewrite or {}).items():
        try:
            rewrite_asts[source] = expr_parse(src_expr)
        except Exception:                                      # noqa: BLE001 — logged, `continue` (:299-303)
            pass
    for it in items:
        def lookup(name, it=it):
            v = ctx_params.get(name, 0.0)
            if v != 0:
                return v
            return it.float_expr_data(name)
        if score_rewrite:
            scores = {}
            for source in score_rewrite:
                a = rewrite_asts.get(source)
                scores[source] = expr_eval(a, lookup) if a is not None else 0.0
            for name, s in scores.items():                     # AddAlgoScores
                it.algo_scores[name] = float(s)
        if ast is not None:
            it.score = expr_eval(ast, lookup)


# ---------------------------------------------------------------------------------------------
# dedup (filter/unique_filter.go:26-49) and response decoders
# ---------------------------------------------------------------------------------------------
def unique_filter(items: List[OracleItem]) -> List[OracleItem]:
    out, seen = [], {}
    for it in items:
        ex = seen.get(it.id)
        if ex is None:
            seen[it.id] = it
            out.append(it)
        else:
            for n, s in it.algo_scores.items():
                ex.add_algo_score(n, s)
            if ex.recall_scores is None:
                ex.recall_scores = {ex.retrieve_id: ex.score}
            ex.recall_scores[it.retrieve_id] = it.score
    return out


def alink_fm_score(prediction_result: float, prediction_score: float) -> float:
    """alinkFMResponse.GetScore (algorithm/eas/fm_response.go:28-34): label 0 → 1 − score."""
    return 1 - prediction_score if prediction_result == 0.0 else prediction_score


def widen_f32(scores_f32: np.ndarray) -> np.ndarray:
    """float32 model outputs widened to float64 AlgoResponse scores
    (algorithm/eas/easyrec_response.go:479-483, tfserving/response.go:51-64)."""
    return np.asarray(scores_f32, dtype=np.float32).astype(np.float64)


def parse_vector_string(s: str) -> np.ndarray:
    """vector_recall.go:70-82: split ' ', keep "i:v" pairs, ParseFloat(v, 32)."""
    out = []
    for vc in s.split(" "):
        if ":" not in vc:
            continue
        vals = vc.split(":")
        if len(vals) == 2:
            try:
                out.append(np.float32(float(vals[1])))
            except ValueError:
                out.append(np.float32(0.0))       # `value, _ :=` ignores the error → 0
    return np.asarray(out, dtype=np.float32)


def recall_cache_string(items: List[OracleItem], recall_name: str) -> str:
    """vector_recall.go:105-110: "id:name:score,..." with Go %v float formatting (shortest repr)."""
    return ",".join("%s:%s:%s" % (it.id, recall_name, go_fmt_float(it.score)) for it in items)


def go_fmt_float(x: float) -> str:
    """fmt %v for float64 = strconv 'g', precision -1: the shortest digits that round-trip, in exponent form for
    exp < -4 || exp >= 6 (strconv/ftoa.go, case 'g': "if precision was the shortest possible, use precision 6 for this decision" —
    the familiar `map[id:1.2345678e+07]` of a JSON number printed with %v; 21 is encoding/json's threshold, not fmt's)."""
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "+Inf" if x > 0 else "-Inf"
    if x == 0:
        return "-0" if math.copysign(1, x) < 0 else "0"
    r = repr(float(x))
    m, _, e = r.partition("e")
    if e:
        exp = int(e)
    else:
        exp = None
    # derive decimal exponent
    digits = m.replace("-", "").replace(".", "").lstrip("0") or "0"
    if exp is None:
        ip = m.lstrip("-").split(".")[0]
        if ip.strip("0") == "":
            frac = m.split(".")[1] if "." in m else ""
            dexp = -(len(frac) - len(frac.lstrip("0"))) - 1
        else:
            dexp = len(ip.lstrip("0")) - 1
    else:
        ip = m.lstrip("-").split(".")[0]
        dexp = exp + (len(ip) - 1)
    digits = digits.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    if dexp < -4 or dexp >= 6:
        mant = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
        return "%s%se%s%02d" % (sign, mant, "+" if dexp >= 0 else "-", abs(dexp))
    if dexp >= 0:
        if len(digits) <= dexp + 1:
            return sign + digits + "0" * (dexp + 1 - len(digits))
        return sign + digits[:dexp + 1] + "." + digits[dexp + 1:]
    return sign + "0." + "0" * (-dexp - 1) + digits


# ---------------------------------------------------------------------------------------------
# EasyrecAlgoDataGenerator  (service/rank/algo_data.go:154-306): per-request feature boxing
# ---------------------------------------------------------------------------------------------
def _go_zero_like(v):
    """feature.defaultValue (algo_data.go:154-171): the Go zero value of the first value's type."""
    if isinstance(v, bool):
        return ""                      # reflect.Bool falls to the default branch: ""
    if isinstance(v, int):
        return 0
    if isinstance(v, float):
        return 0.0
    return ""


class EasyrecGenerator:
    """Restatement of EasyrecAlgoDataGenerator: AddFeatures keeps one list per feature name and fills an
    item that lacks the feature with the column's zero value; GeneratorAlgoData emits and resets them."""

    def __init__(self, context_features):
        self.items = []
        self.context = {}
        self.parse_feature = True                      # :187 — the schema is the configured list
        self.item_features = [(n, "") for n in context_features]       # typed string → default ""
        self.user = {}
        self.input_features = []
        self.input_map = None
        self.parse_input = False

    def set_item_features(self, names):                # :204-221
        if names:
            self.input_map = {}
            if names[0] != "*":
                self.parse_input = True
                self.input_features = [(n, "") for n in names]
        else:
            self.parse_input = True

    def add_features(self, item_id, item_features, user_features):     # :223-271
        self.items.append(item_id)
        if not self.parse_feature:
            self.item_features = [(k, _go_zero_like(v)) for k, v in item_features.items()]
            self.user = user_features
            self.parse_feature = True
        if not self.parse_input:
            ctx_names = [n for n, _ in self.item_features]
            for k, v in item_features.items():
                if k not in ctx_names:
                    self.input_features.append((k, _go_zero_like(v)))
            self.parse_input = True
        if not self.user:
            self.user = user_features
        for n, d in self.item_features:
            self.context.setdefault(n, []).append(item_features.get(n, d))
        if self.input_map is not None:
            for n, d in self.input_features:
                self.input_map.setdefault(n, []).append(item_features.get(n, d))

    def generate(self):                                # :273-302
        out = {"user_features": dict(self.user), "item_ids": list(self.items),
               "context_features": {k: list(v) for k, v in self.context.items()},
               "item_features": {k: list(v) for k, v in (self.input_map or {}).items()}}
        for v in self.context.values():
            del v[:]
        for v in (self.input_map or {}).values():
            del v[:]
        self.items = []
        return out
