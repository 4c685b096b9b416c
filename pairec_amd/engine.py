"""Thin object layer over the C ABI (include/pairec_gpu.h): Context, Table, RankModel, Expr.

Host arrays are numpy; "dev" methods take raw device addresses (ints), e.g. torch tensors'
`.data_ptr()` — torch is used by callers only as plumbing for device memory and RCCL.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib

PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
MODEL_DNN3, MODEL_FM_TWOTOWER, MODEL_DNN3_MULTI = 1, 2, 3
MAX_QUERIES = 256         # per table pass (32 when dim > 128)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.L = _lib.load()
        if stream is not None and int(stream) == 0:
            # handle 0 is HIP's null stream (torch's default stream): pg_init reads NULL as "create a private
            # stream", which would silently leave the caller's torch ops and the library's kernels unordered
            raise ValueError("Context: cannot adopt the null stream (handle 0); pass a dedicated stream's handle "
                             "(torch.cuda.Stream().cuda_stream) or None for a private one")
        h = C.c_void_p()
        _lib.check(self.L.pg_init(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        self.stream_handle = int(stream) if stream else None
        self.torch_stream = None

    def close(self):
        if self.h:
            self.L.pg_shutdown(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        _lib.check(self.L.pg_synchronize(self.h))

    def debug_stall(self, ms: int) -> None:
        """Occupy this context's stream for `ms` milliseconds (test aid for the deadline paths)."""
        _lib.check(self.L.pg_debug_stall(self.h, int(ms)))

    def set_option(self, name: str, value) -> None:
        """Developer / test knob of this context (pg_set_option)."""
        _lib.check(self.L.pg_set_option(self.h, name.encode(), str(value).encode()))

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _lib.check(self.L.pg_device_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, p: int):
        _lib.check(self.L.pg_device_free(self.h, C.c_void_p(p)))

    def h2d(self, dst: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        _lib.check(self.L.pg_memcpy_h2d(self.h, C.c_void_p(dst), _ptr(a), a.nbytes))

    def d2h(self, a: np.ndarray, src: int):
        assert a.flags.c_contiguous
        _lib.check(self.L.pg_memcpy_d2h(self.h, _ptr(a), C.c_void_p(src), a.nbytes))

    def to_device(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        p = self.malloc(max(a.nbytes, 16))
        self.h2d(p, a)
        return p

    def stats(self) -> _lib.PgStats:
        s = _lib.PgStats()
        _lib.check(self.L.pg_stats(self.h, C.byref(s)))
        return s

    def last_scan_kernel(self) -> Tuple[float, int]:
        ms, b = C.c_double(), C.c_uint64()
        _lib.check(self.L.pg_last_scan_kernel_ms(self.h, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    # ---- sort / expr (context-level ops) ----------------------------------------------------
    def sort_scores(self, scores: np.ndarray, seg_offsets: Optional[Sequence[int]] = None,
                    descending: bool = True) -> np.ndarray:
        s = np.ascontiguousarray(scores, dtype=np.float64)
        if seg_offsets is None:
            seg_offsets = [0, s.shape[0]]
        so = np.ascontiguousarray(seg_offsets, dtype=np.uint32)
        if so.shape[0] < 1 or int(so[-1]) != s.shape[0]:
            # (the C call takes the item count from the last offset: the buffers must be that long)
            raise ValueError("sort_scores: seg_offsets must end at len(scores) = %d" % s.shape[0])
        out = np.zeros(s.shape[0], dtype=np.uint32)
        _lib.check(self.L.pg_sort_scores(self.h, _ptr(s), _ptr(so), so.shape[0] - 1,
                                         int(descending), _ptr(out)))
        return out


class Table:
    """HBM-resident embedding table (module.VectorDao replacement)."""

    def __init__(self, ctx: Context, rows: int, dim: int, row_offset: int = 0):
        self.ctx, self.rows, self.dim, self.row_offset = ctx, rows, dim, row_offset
        h = C.c_void_p()
        _lib.check(ctx.L.pg_table_create(ctx.h, rows, dim, row_offset, C.byref(h)))
        self.h = h

    def destroy(self):
        if self.h:
            _lib.check(self.ctx.L.pg_table_destroy(self.ctx.h, self.h))
            self.h = None

    def hbm_read_probe(self, reps: int = 3) -> float:
        """Measured streaming-read rate over this table's rows, GB/s."""
        g = C.c_double()
        _lib.check(self.ctx.L.pg_hbm_read_probe(self.ctx.h, self.h, reps, C.byref(g)))
        return g.value

    def screen_info(self) -> Tuple[int, float, float]:
        """(element bytes of the shadow the screened recall streams — 1 int8, 2 bf16, 0 exact fp32 scan —,
        int8 scale, int8 max row residual); builds the shadow if it is not built yet."""
        eb, sc, rs = C.c_int(), C.c_float(), C.c_float()
        _lib.check(self.ctx.L.pg_table_screen_info(self.ctx.h, self.h, C.byref(eb), C.byref(sc), C.byref(rs)))
        return eb.value, sc.value, rs.value

    def fill_synthetic(self, seed: int, normalize: bool = True):
        _lib.check(self.ctx.L.pg_table_fill_synthetic(self.ctx.h, self.h, seed, int(normalize)))

    def fill_gaussian(self, seed: int, sigma: float = 1.0):
        _lib.check(self.ctx.L.pg_table_fill_gaussian(self.ctx.h, self.h, seed, float(sigma)))

    def fill_mixture(self, seed: int, n_centres: int, sigma: float):
        """clustered rows: n_centres centres on the unit sphere, within-cluster noise of norm ~ sigma, normalised"""
        _lib.check(self.ctx.L.pg_table_fill_mixture(self.ctx.h, self.h, seed, int(n_centres), float(sigma)))

    def upload(self, rows: np.ndarray, row0: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.dim
        _lib.check(self.ctx.L.pg_table_upload(self.ctx.h, self.h, row0, rows.shape[0], _ptr(rows)))

    def download(self, row0: int, nrows: int) -> np.ndarray:
        out = np.empty((nrows, self.dim), dtype=np.float32)
        _lib.check(self.ctx.L.pg_table_download(self.ctx.h, self.h, row0, nrows, _ptr(out)))
        return out

    def gather(self, rows: Sequence[int]) -> np.ndarray:
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.empty((r.shape[0], self.dim), dtype=np.float32)
        _lib.check(self.ctx.L.pg_table_gather(self.ctx.h, self.h, _ptr(r), r.shape[0], _ptr(out)))
        return out

    def swap(self, other: "Table"):
        _lib.check(self.ctx.L.pg_table_swap(self.ctx.h, self.h, other.h))
        self.row_offset, other.row_offset = other.row_offset, self.row_offset

    def recall_topk(self, queries: np.ndarray, k: int):
        """queries [nq][dim] → (rows [nq][k] uint64 global ids, scores [nq][k] f32, counts [nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        rows = np.empty((nq, k), dtype=np.uint64)
        scores = np.empty((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        per_pass = MAX_QUERIES if self.dim <= 128 else 32
        for s in range(0, nq, per_pass):             # one table pass per batch of queries
            e = min(nq, s + per_pass)
            r_, s_, c_ = rows[s:e], scores[s:e], counts[s:e]
            _lib.check(self.ctx.L.pg_recall_topk(self.ctx.h, self.h, _ptr(q[s:e]), e - s, k,
                                                 _ptr(r_), _ptr(s_), _ptr(c_)))
        return rows, scores, counts

    def recall_topk_l2(self, queries: np.ndarray, k: int):
        """HologresVectorRecallV2: queries [nq][dim] → (rows [nq][k], squared Euclidean distances [nq][k] ascending, counts)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        rows = np.empty((nq, k), dtype=np.uint64)
        dist = np.empty((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        for s in range(0, nq, MAX_QUERIES):
            e = min(nq, s + MAX_QUERIES)
            r_, d_, c_ = rows[s:e], dist[s:e], counts[s:e]
            _lib.check(self.ctx.L.pg_recall_topk_l2(self.ctx.h, self.h, _ptr(q[s:e]), e - s, k, _ptr(r_), _ptr(d_), _ptr(c_)))
        return rows, dist, counts

    def recall_topk_where(self, feats: "Features", column: str, op: str, value: int, queries: np.ndarray, k: int, l2: bool = False):
        """A Hologres vector recall with its WhereClause `column OP value` (op in > >= < <= == !=): only rows that pass are
        candidates.  → (rows, scores or distances, counts)."""
        ops = {">": 0, ">=": 1, "<": 2, "<=": 3, "==": 4, "!=": 5}
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        rows = np.empty((nq, k), dtype=np.uint64)
        scores = np.empty((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        col = self.ctx.L.pg_features_column_index(feats.h, column.encode())
        for s in range(0, nq, MAX_QUERIES):
            e = min(nq, s + MAX_QUERIES)
            r_, s_, c_ = rows[s:e], scores[s:e], counts[s:e]
            _lib.check(self.ctx.L.pg_recall_topk_where(self.ctx.h, self.h, feats.h, col, ops[op], int(value), 1 if l2 else 0,
                                                       _ptr(q[s:e]), e - s, k, _ptr(r_), _ptr(s_), _ptr(c_)))
        return rows, scores, counts

    def view(self, feats: "Features", column: str, op: str, value: int) -> "Table":
        """The rows `column OP value` admits, as a table of their own whose recalls answer with THIS table's row ids
        (pg_table_view_create): the WhereClause of a Hologres recall whose constant is fixed when the recall is built."""
        ops = {">": 0, ">=": 1, "<": 2, "<=": 3, "==": 4, "!=": 5}
        col = self.ctx.L.pg_features_column_index(feats.h, column.encode())
        h = C.c_void_p()
        _lib.check(self.ctx.L.pg_table_view_create(self.ctx.h, self.h, feats.h, col, ops[op], int(value), C.byref(h)))
        v = Table.__new__(Table)
        v.ctx, v.dim, v.row_offset, v.h = self.ctx, self.dim, 0, h
        rows = C.c_uint64()
        _lib.check(self.ctx.L.pg_table_info(h, C.byref(rows), None, None))
        v.rows = rows.value
        return v

    def i2i_recall(self, trigger_rows, k: int, trigger_table: Optional["Table"] = None):
        """I2IVectorRecall: rows of `trigger_table` (default: this table) are the queries."""
        tr = np.ascontiguousarray(trigger_rows, dtype=np.uint32)
        n = tr.shape[0]
        rows = np.empty((n, k), dtype=np.uint64)
        scores = np.empty((n, k), dtype=np.float32)
        counts = np.zeros(n, dtype=np.uint32)
        _lib.check(self.ctx.L.pg_i2i_recall(self.ctx.h, (trigger_table or self).h, _ptr(tr), n, self.h, k,
                                            _ptr(rows), _ptr(scores), _ptr(counts)))
        return rows, scores, counts

    def recall_topk_dev(self, d_queries: int, nq: int, k: int, d_out_rows: int, d_out_scores: int):
        counts = np.zeros(nq, dtype=np.uint32)
        _lib.check(self.ctx.L.pg_recall_topk_dev(self.ctx.h, self.h, C.c_void_p(d_queries), nq, k,
                                                 C.c_void_p(d_out_rows), C.c_void_p(d_out_scores),
                                                 _ptr(counts)))
        return counts


def pack_dnn3(w1, b1, w2, b2, w3, b3, d_user: int) -> bytes:
    w1 = np.ascontiguousarray(w1, dtype=np.float32)
    w2 = np.ascontiguousarray(w2, dtype=np.float32)
    din, h1 = w1.shape
    h2 = w2.shape[1]
    return (struct.pack("<4I", d_user, din - d_user, h1, h2) + w1.tobytes() +
            np.ascontiguousarray(b1, dtype=np.float32).tobytes() + w2.tobytes() +
            np.ascontiguousarray(b2, dtype=np.float32).tobytes() +
            np.ascontiguousarray(w3, dtype=np.float32).tobytes() + struct.pack("<f", float(b3)))


def pack_dnn3_multi(w1, b1, w2, b2, w3, b3, d_user: int) -> bytes:
    """PG_MODEL_DNN3_MULTI blob (include/pairec_gpu.h): w3 [h2][n_out], b3 [n_out] — n_out heads on one trunk."""
    w1 = np.ascontiguousarray(w1, dtype=np.float32)
    w2 = np.ascontiguousarray(w2, dtype=np.float32)
    w3 = np.ascontiguousarray(w3, dtype=np.float32)
    b3 = np.ascontiguousarray(b3, dtype=np.float32).reshape(-1)
    din, h1 = w1.shape
    h2 = w2.shape[1]
    assert w3.shape == (h2, b3.shape[0])
    return (struct.pack("<5I", d_user, din - d_user, h1, h2, b3.shape[0]) + w1.tobytes() +
            np.ascontiguousarray(b1, dtype=np.float32).tobytes() + w2.tobytes() +
            np.ascontiguousarray(b2, dtype=np.float32).tobytes() + w3.tobytes() + b3.tobytes())


def pack_fm2t(w) -> bytes:
    """w: object with the attributes of oracle.Fm2tWeights (duck-typed; no oracle import here)."""
    parts = [struct.pack("<7If", w.nuf, w.nif, w.k, w.d_user, w.t_h1, w.t_out, w.vocab, w.fm_b)]
    for a in (w.uw1, w.ub1, w.uw2, w.ub2, w.iw1, w.ib1, w.iw2, w.ib2):
        parts.append(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    for f in range(w.nuf + w.nif):
        parts.append(np.ascontiguousarray(w.field_emb[f], dtype=np.float32).tobytes())
        parts.append(np.ascontiguousarray(w.field_lin[f], dtype=np.float32).tobytes())
    return b"".join(parts)


class RankModel:
    """Rank model resident in HBM (algorithm/eas | tfserving predict replacement)."""

    def __init__(self, ctx: Context, kind: int, prec: int, blob: bytes):
        self.ctx, self.kind, self.prec = ctx, kind, prec
        self._blob_head = bytes(blob[:28])
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _lib.check(ctx.L.pg_model_load(ctx.h, kind, prec, buf, len(blob), C.byref(h)))
        self.h = h
        n = C.c_uint32()
        _lib.check(ctx.L.pg_model_num_outputs(h, C.byref(n)))
        self.n_out = n.value

    def destroy(self):
        if self.h:
            _lib.check(self.ctx.L.pg_model_destroy(self.ctx.h, self.h))
            self.h = None

    def rank_dnn3(self, table: Table, user_vecs: np.ndarray, cand_rows: np.ndarray,
                  req_offsets: Sequence[int]) -> np.ndarray:
        """scores [n_items]; a multi-output model: [n_out][n_items] (one plane per head)."""
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        ro = np.ascontiguousarray(req_offsets, dtype=np.uint32)
        out = np.empty((self.n_out, c.shape[0]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_rank_dnn3(self.ctx.h, self.h, table.h, _ptr(u), _ptr(c), _ptr(ro),
                                           ro.shape[0] - 1, _ptr(out)))
        return out[0] if self.n_out == 1 else out

    def rank_dnn3_dev(self, table: Table, d_user_vecs: int, d_cand_rows: int, d_req_offsets: int,
                      n_req: int, n_items: int, d_out: int):
        _lib.check(self.ctx.L.pg_rank_dnn3_dev(self.ctx.h, self.h, table.h, C.c_void_p(d_user_vecs),
                                               C.c_void_p(d_cand_rows), C.c_void_p(d_req_offsets),
                                               n_req, n_items, C.c_void_p(d_out)))

    def rank_fm2t(self, user_vecs, user_field_ids, item_field_ids, req_offsets) -> np.ndarray:
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
        itf = np.ascontiguousarray(item_field_ids, dtype=np.int32)
        ro = np.ascontiguousarray(req_offsets, dtype=np.uint32)
        out = np.empty(int(ro[-1]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_rank_fm2t(self.ctx.h, self.h, _ptr(u), _ptr(uf), _ptr(itf),
                                           _ptr(ro), ro.shape[0] - 1, _ptr(out)))
        return out


    def user_embedding(self, user_vecs: np.ndarray) -> np.ndarray:
        """Two-tower user embedding [n][t_out] (pg_fm2t_user_embedding)."""
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        u = u.reshape(-1, u.shape[-1])
        hdr = struct.unpack("<7I", self._blob_head)
        out = np.empty((u.shape[0], hdr[5]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_fm2t_user_embedding(self.ctx.h, self.h, _ptr(u), u.shape[0], _ptr(out)))
        return out

    def online_vector_recall(self, item_emb: Table, user_vecs: np.ndarray, k: int):
        """OnlineVectorRecall: user tower → top-k of the item-embedding table."""
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        u = u.reshape(-1, u.shape[-1])
        n = u.shape[0]
        rows = np.empty((n, k), dtype=np.uint64)
        scores = np.empty((n, k), dtype=np.float32)
        counts = np.zeros(n, dtype=np.uint32)
        _lib.check(self.ctx.L.pg_online_vector_recall(self.ctx.h, self.h, item_emb.h, _ptr(u), n, k, _ptr(rows),
                                                      _ptr(scores), _ptr(counts)))
        return rows, scores, counts

    def rank_fm2t_rows(self, feats: "Features", item_field_names, user_vecs, user_field_ids, cand_rows,
                       req_offsets) -> np.ndarray:
        """FM + two-tower rank from candidate rows: the item field ids come from feature columns."""
        ctx = self.ctx
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
        cr = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        ro = np.ascontiguousarray(req_offsets, dtype=np.uint32)
        cols = feats._cols(item_field_names)
        n = int(ro[-1])
        d_u, d_uf, d_cr, d_ro = ctx.to_device(u), ctx.to_device(uf), ctx.to_device(cr), ctx.to_device(ro)
        d_o = ctx.malloc(max(n * 4, 16))
        _lib.check(ctx.L.pg_rank_fm2t_rows_dev(ctx.h, self.h, feats.h, _ptr(cols), d_u, d_uf, d_cr, d_ro,
                                               ro.shape[0] - 1, n, d_o))
        out = np.zeros(n, dtype=np.float32)
        ctx.d2h(out, d_o)
        for p in (d_u, d_uf, d_cr, d_ro, d_o):
            ctx.free(p)
        return out


class ItemRows:
    """Materialised item records of an FM + two-tower model (pg_fm2t_item_rows_*): one contiguous 640-B record per
    item row, built once from the model's field tables and the item-field columns."""

    def __init__(self, model: "RankModel", feats: "Features", item_field_names):
        self.ctx, self.model, self.feats = model.ctx, model, feats
        cols = feats._cols(item_field_names)
        h = C.c_void_p()
        _lib.check(self.ctx.L.pg_fm2t_item_rows_build(self.ctx.h, model.h, feats.h, _ptr(cols), C.byref(h)))
        self.h = h

    def update(self, row0: int, nrows: int):
        _lib.check(self.ctx.L.pg_fm2t_item_rows_update(self.ctx.h, self.h, row0, nrows))

    def destroy(self):
        if self.h:
            _lib.check(self.ctx.L.pg_fm2t_item_rows_destroy(self.ctx.h, self.h))
            self.h = None

    def rank(self, user_vecs, user_field_ids, cand_rows, req_offsets) -> np.ndarray:
        """pg_rank_fm2t_irows (host buffers)."""
        u = np.ascontiguousarray(user_vecs, dtype=np.float32)
        uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
        cr = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        ro = np.ascontiguousarray(req_offsets, dtype=np.uint32)
        out = np.empty(int(ro[-1]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_rank_fm2t_irows(self.ctx.h, self.model.h, self.h, _ptr(u), _ptr(uf), _ptr(cr), _ptr(ro),
                                                 ro.shape[0] - 1, _ptr(out)))
        return out


def rank_fm2t_rows_host(model: "RankModel", feats: "Features", item_field_names, user_vecs, user_field_ids, cand_rows,
                        req_offsets) -> np.ndarray:
    """pg_rank_fm2t_rows: the host-buffer form (what a caller-made batch of IAlgorithm.Run calls passes)."""
    ctx = model.ctx
    u = np.ascontiguousarray(user_vecs, dtype=np.float32)
    uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
    cr = np.ascontiguousarray(cand_rows, dtype=np.uint32)
    ro = np.ascontiguousarray(req_offsets, dtype=np.uint32)
    cols = feats._cols(item_field_names)
    out = np.empty(int(ro[-1]), dtype=np.float32)
    _lib.check(ctx.L.pg_rank_fm2t_rows(ctx.h, model.h, feats.h, _ptr(cols), _ptr(u), _ptr(uf), _ptr(cr), _ptr(ro),
                                       ro.shape[0] - 1, _ptr(out)))
    return out


class Expr:
    """Compiled RankConfig.RankScore expression (utils/ast replacement)."""

    def __init__(self, source: str, ast_type: str = ""):
        """ast_type "antlr": the subset of the reference's second evaluator (pg_expr_compile_typed)."""
        self.L = _lib.load()
        h = C.c_void_p()
        if ast_type:
            _lib.check(self.L.pg_expr_compile_typed(source.encode("utf-8"), ast_type.encode("utf-8"), C.byref(h)))
        else:
            _lib.check(self.L.pg_expr_compile(source.encode("utf-8"), C.byref(h)))
        self.h = h
        n = self.L.pg_expr_num_vars(h)
        self.var_names = [self.L.pg_expr_var_name(h, i).decode("utf-8") for i in range(n)]

    def free(self):
        if self.h:
            self.L.pg_expr_free(self.h)
            self.h = None

    def set_score_rewrites(self, rewrites: dict):
        """RankConfig.ScoreRewrite {source: expression} of the scene this RankScore belongs to (pg_expr_set_score_rewrites).
        An expression that does not compile is passed as NULL — the reference scores such a source 0."""
        names = list(rewrites.keys())
        exprs = []
        for nm in names:
            try:
                exprs.append(Expr(rewrites[nm]))
            except _lib.PgError:
                exprs.append(None)
        arr_n = (C.c_char_p * max(len(names), 1))(*[nm.encode("utf-8") for nm in names])
        arr_e = (C.c_void_p * max(len(names), 1))(*[(x.h if x is not None else None) for x in exprs])
        try:
            _lib.check(self.L.pg_expr_set_score_rewrites(self.h, len(names), arr_n, arr_e))
        finally:
            for x in exprs:
                if x is not None:
                    x.free()

    def eval(self, ctx: Context, vars_: np.ndarray) -> np.ndarray:
        """vars_: [n_vars][n_items] fp64 in var_names order → fused scores [n_items] fp64."""
        v = np.ascontiguousarray(vars_, dtype=np.float64).reshape(len(self.var_names), -1) \
            if len(self.var_names) else np.zeros((0, int(np.shape(vars_)[-1])), dtype=np.float64)
        n = v.shape[1]
        out = np.empty(n, dtype=np.float64)
        _lib.check(self.L.pg_expr_eval(ctx.h, self.h, _ptr(v) if v.size else None, n, _ptr(out)))
        return out


def recommend_dnn3(ctx: Context, table: Table, model: "RankModel", expr: "Expr", rank_var: str, queries: np.ndarray,
                   k: int):
    """pg_recommend_dnn3_dev on host arrays (tests, tools): queries [R][dim] →
    rows [R][k] u64, recall scores [R][k] f32, model scores [R][k] f32, fused [R][k] f64, order [R][k] u32, counts [R]."""
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, table.dim)
    R = q.shape[0]
    n = R * k
    d_q = ctx.to_device(q)
    bufs = [ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 8), ctx.malloc(n * 4),
            ctx.malloc(max(R * 4, 16))]
    try:
        _lib.check(ctx.L.pg_recommend_dnn3_dev(ctx.h, table.h, model.h, expr.h, rank_var.encode(), d_q, R, k, *bufs))
        outs = [np.zeros((R, k), np.uint64), np.zeros((R, k), np.float32), np.zeros((R, k), np.float32),
                np.zeros((R, k), np.float64), np.zeros((R, k), np.uint32), np.zeros(R, np.uint32)]
        for a, p_ in zip(outs, bufs):
            ctx.d2h(a, p_)
    finally:
        for p_ in [d_q] + bufs:
            ctx.free(p_)
    return tuple(outs)


class Coalescer:
    """Cross-request batching (pg_coalescer_*): every method serves ONE request and may be called from any number
    of threads at once (ctypes releases the GIL for the duration of the call); the library forms the batches.

    Two ways to build one: the single-DNN form (model / expr / rank_var: pg_coalescer_create), or a scene
    (`algos` = [(name, RankModel) or (name, RankModel, Features, item field column names)], expr, `dpp` =
    {"candidates": C, "alpha": a, "window": w, "normalize_emb": True, "norm_relevance_score": 0},
    query_model, trigger_table: pg_coalescer_create_scene)."""

    def __init__(self, ctx: Context, table: Table, k: int, model: Optional["RankModel"] = None,
                 expr: Optional["Expr"] = None, rank_var: str = "", max_batch: int = 0, max_wait_us: int = 0,
                 depth: int = 0, max_top_n: int = 0, max_rank_items: int = 0, timeout_us: int = 0,
                 algos=None, dpp: Optional[dict] = None, query_model: Optional["RankModel"] = None,
                 trigger_table: Optional[Table] = None, max_rerank_items: int = 0, max_hook_dim: int = 0):
        self.ctx, self.table, self.k = ctx, table, k
        self.max_top_n = max_top_n or k
        cfg = _lib.PgCoalescerConfig(k, max_batch, max_wait_us, depth, max_top_n, max_rank_items, timeout_us)
        h = C.c_void_p()
        scene = algos is not None or dpp is not None or query_model is not None or trigger_table is not None \
            or max_rerank_items or max_hook_dim
        self.n_algos = 1 if model else 0
        self.n_planes = getattr(model, "n_out", 1) if model else 0
        self.algo_outputs = [self.n_planes] if model else []
        self.dnn_heads = self.n_planes or 1
        if not scene:
            _lib.check(ctx.L.pg_coalescer_create(ctx.h, table.h, model.h if model else None, expr.h if expr else None,
                                                 rank_var.encode() if rank_var else None, C.byref(cfg), C.byref(h)))
        else:
            if algos is None:
                algos = [(rank_var, model)] if model else []
            arr = (_lib.PgRankAlgo * max(len(algos), 1))()
            self._keep = []
            for i, a in enumerate(algos):
                name, m = a[0], a[1]
                nm = name.encode()
                self._keep.append(nm)
                arr[i].model = m.h
                arr[i].name = nm
                if len(a) == 3 and isinstance(a[2], (list, tuple)):     # (name, multi-output model, output names)
                    onames = (C.c_char_p * len(a[2]))(*[str(x).encode() for x in a[2]])
                    self._keep.append(onames)
                    arr[i].output_names = onames
                elif len(a) == 3:                     # (name, model, ItemRows)
                    arr[i].item_rows = a[2].h
                elif len(a) > 3:                      # (name, model, Features, item field column names)
                    feats, cols = a[2], a[3]
                    idx = feats._cols(cols)
                    carr = (C.c_int32 * len(idx))(*[int(x) for x in idx])
                    self._keep.append(carr)
                    arr[i].features = feats.h
                    arr[i].item_field_cols = carr
            sc = _lib.PgSceneConfig()
            sc.base = cfg
            sc.algos = arr
            sc.n_algos = len(algos)
            sc.rank_score = expr.h if expr else None
            if dpp is not None:
                sc.rerank = 1
                sc.rerank_candidates = int(dpp["candidates"])
                sc.dpp = _lib.PgDppOptions(float(dpp.get("alpha", 1.0)), 0, int(dpp.get("window", 10)),
                                           int(dpp.get("normalize_emb", True)), 1,
                                           int(dpp.get("norm_relevance_score", 0)), 1, 0)
            sc.query_model = query_model.h if query_model else None
            sc.trigger_table = trigger_table.h if trigger_table else None
            sc.max_rerank_items = max_rerank_items
            sc.max_hook_dim = max_hook_dim
            self.n_algos = len(algos)
            self.n_planes = sum(getattr(a[1], "n_out", 1) for a in algos)
            self.algo_outputs = [getattr(a[1], "n_out", 1) for a in algos]
            self.dnn_heads = next((a[1].n_out for a in algos if a[1].kind in (MODEL_DNN3, MODEL_DNN3_MULTI)), 1)
            _lib.check(ctx.L.pg_coalescer_create_scene(ctx.h, table.h, C.byref(sc), C.byref(h)))
        self.h = h

    def destroy(self):
        if self.h:
            _lib.check(self.ctx.L.pg_coalescer_destroy(self.h))
            self.h = None

    def _recall_out(self):
        return np.empty(self.k, dtype=np.uint64), np.empty(self.k, dtype=np.float32), C.c_uint32()

    def recall(self, query: np.ndarray):
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.table.dim)
        rows, scores, cnt = self._recall_out()
        _lib.check(self.ctx.L.pg_coalescer_recall(self.h, _ptr(q), _ptr(rows), _ptr(scores), C.byref(cnt)))
        return rows, scores, cnt.value

    def recall_l2(self, query: np.ndarray):
        """HologresVectorRecallV2: one request; (rows, squared Euclidean distances ascending, count)."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.table.dim)
        rows, dist, cnt = self._recall_out()
        _lib.check(self.ctx.L.pg_coalescer_recall_l2(self.h, _ptr(q), _ptr(rows), _ptr(dist), C.byref(cnt)))
        return rows, dist, cnt.value

    def i2i_recall(self, trigger_row: int):
        rows, scores, cnt = self._recall_out()
        _lib.check(self.ctx.L.pg_coalescer_i2i_recall(self.h, int(trigger_row), _ptr(rows), _ptr(scores), C.byref(cnt)))
        return rows, scores, cnt.value

    def online_recall(self, user_vec: np.ndarray):
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(-1)
        rows, scores, cnt = self._recall_out()
        _lib.check(self.ctx.L.pg_coalescer_online_recall(self.h, _ptr(u), _ptr(rows), _ptr(scores), C.byref(cnt)))
        return rows, scores, cnt.value

    def rank_dnn3(self, user_vec: np.ndarray, cand_rows: np.ndarray) -> np.ndarray:
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(-1)
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        heads = getattr(self, "dnn_heads", 1)
        out = np.empty((heads, c.shape[0]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_coalescer_rank_dnn3(self.h, _ptr(u), _ptr(c), c.shape[0], _ptr(out)))
        return out[0] if heads == 1 else out

    def rank(self, algo: int, user_vec: np.ndarray, cand_rows: np.ndarray, user_field_ids=None) -> np.ndarray:
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(-1)
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        uf = None if user_field_ids is None else np.ascontiguousarray(user_field_ids, dtype=np.int32)
        heads = self.algo_outputs[algo]
        out = np.empty((heads, c.shape[0]), dtype=np.float32)
        _lib.check(self.ctx.L.pg_coalescer_rank(self.h, algo, _ptr(u), _ptr(uf) if uf is not None else None, _ptr(c),
                                                c.shape[0], _ptr(out)))
        return out[0] if heads == 1 else out

    def rank_fm2t(self, user_vec: np.ndarray, user_field_ids, cand_rows: np.ndarray) -> np.ndarray:
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(-1)
        uf = np.ascontiguousarray(user_field_ids, dtype=np.int32)
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        out = np.empty(c.shape[0], dtype=np.float32)
        _lib.check(self.ctx.L.pg_coalescer_rank_fm2t(self.h, _ptr(u), _ptr(uf), _ptr(c), c.shape[0], _ptr(out)))
        return out

    def recommend(self, user_vec: np.ndarray, top_n: int, user_field_ids=None):
        """→ (rows, recall scores, model scores, fused scores) of the page (the first top_n entries of the sorted list,
        or DPPSort's picks when the scene has the stage), count.  With several rank algorithms (or user_field_ids)
        the model scores are [n_algos][top_n]."""
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(self.table.dim)
        rows = np.empty(top_n, dtype=np.uint64)
        rec = np.empty(top_n, dtype=np.float32)
        fus = np.empty(top_n, dtype=np.float64)
        cnt = C.c_uint32()
        if user_field_ids is None and self.n_planes <= 1:
            rnk = np.empty(top_n, dtype=np.float32)
            _lib.check(self.ctx.L.pg_coalescer_recommend(self.h, _ptr(u), top_n, _ptr(rows), _ptr(rec), _ptr(rnk),
                                                         _ptr(fus), C.byref(cnt)))
        else:
            rnk = np.empty((self.n_planes, top_n), dtype=np.float32)
            uf = None if user_field_ids is None else np.ascontiguousarray(user_field_ids, dtype=np.int32)
            _lib.check(self.ctx.L.pg_coalescer_recommend_ex(self.h, _ptr(u), _ptr(uf) if uf is not None else None, top_n,
                                                            _ptr(rows), _ptr(rec), _ptr(rnk), _ptr(fus), C.byref(cnt)))
        return rows, rec, rnk, fus, cnt.value

    def dpp(self, cand_rows, rel, alpha: float, topn: int, window: int, normalize_emb: bool = True,
            ensure_pos_similarity: bool = True, norm_relevance_score: int = 0, hook_emb: Optional[np.ndarray] = None,
            has_table: bool = True):
        """pg_coalescer_dpp: one request's DPPSort; → (picked indices, relevance scores as used)."""
        r = np.ascontiguousarray(rel, dtype=np.float64)
        n = r.shape[0]
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32) if has_table else None
        hk = None if hook_emb is None else np.ascontiguousarray(hook_emb, dtype=np.float64).reshape(n, -1)
        opt = _lib.PgDppOptions(alpha, topn, window, int(normalize_emb), int(ensure_pos_similarity),
                                int(norm_relevance_score), int(has_table), 0 if hk is None else hk.shape[1])
        out = np.zeros(max(topn, 1), dtype=np.uint32)
        used = np.zeros(max(n, 1), dtype=np.float64)
        cnt = C.c_uint32()
        _lib.check(self.ctx.L.pg_coalescer_dpp(self.h, _ptr(c) if c is not None else None, _ptr(r), n, C.byref(opt),
                                               _ptr(hk) if hk is not None else None, _ptr(out), C.byref(cnt), _ptr(used)))
        return out[:cnt.value], used[:n]

    def ssd(self, cand_rows, rel, gamma: float, topn: int, window: int, normalize_emb: bool = True,
            ensure_pos_similarity: bool = True, norm_quality_score: int = 0, use_ssd_star: bool = False):
        """pg_coalescer_ssd: one request's SSDSort; → (picked indices, quality scores)."""
        c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
        r = np.ascontiguousarray(rel, dtype=np.float64)
        out = np.zeros(max(c.shape[0], 1), dtype=np.uint32)
        qual = np.zeros(max(c.shape[0], 1), dtype=np.float64)
        cnt = C.c_uint32()
        _lib.check(self.ctx.L.pg_coalescer_ssd(self.h, _ptr(c), _ptr(r), c.shape[0], gamma, topn, window, int(normalize_emb),
                                               int(ensure_pos_similarity), int(norm_quality_score), int(use_ssd_star),
                                               _ptr(out), C.byref(cnt), _ptr(qual)))
        return out[:cnt.value], qual[:c.shape[0]]

    def stats(self) -> _lib.PgCoalescerStats:
        s = _lib.PgCoalescerStats()
        _lib.check(self.ctx.L.pg_coalescer_stats(self.h, C.byref(s)))
        return s


class GroupCoalescer(Coalescer):
    """pg_coalescer_create_group: single-request recommend calls batched into steps of a shard group."""

    def __init__(self, group: "ShardGroup", expr: "Expr", rank_var: str, k: int, max_top_n: int = 0, dpp_candidates: int = 0,
                 dpp_alpha: float = 1.0, dpp_window: int = 10, dpp_normalize_emb: bool = True, max_batch: int = 0,
                 max_wait_us: int = 0, depth: int = 0, timeout_us: int = 0):
        self.L = group.L
        self.group, self.k, self.n_algos = group, k, 1
        self.n_planes, self.algo_outputs, self.dnn_heads = 1, [1], 1
        self.max_top_n = max_top_n or k
        plan = _lib.PgGroupPlan(k, dpp_candidates, dpp_alpha, dpp_window, int(dpp_normalize_emb))
        cfg = _lib.PgCoalescerConfig(k, max_batch, max_wait_us, depth, max_top_n, 0, timeout_us)
        h = C.c_void_p()
        _lib.check(self.L.pg_coalescer_create_group(group.h, expr.h, rank_var.encode(), C.byref(plan), C.byref(cfg), C.byref(h)))
        self.h = h

        class _T:          # (what Coalescer.recommend reads off its table / context)
            dim = group.dim
        self.table = _T()

        class _Cx:
            L = group.L
        self.ctx = _Cx()


class Router:
    """pg_router_*: per-request calls spread over replica coalescers (least outstanding requests first)."""

    def __init__(self, replicas: Sequence["Coalescer"]):
        self.L = _lib.load()
        self.replicas = list(replicas)
        arr = (C.c_void_p * len(replicas))(*[r.h for r in replicas])
        h = C.c_void_p()
        _lib.check(self.L.pg_router_create(arr, len(replicas), C.byref(h)))
        self.h = h
        self.dim = replicas[0].table.dim
        self.k = replicas[0].k

    def destroy(self):
        if self.h:
            self.L.pg_router_destroy(self.h)
            self.h = None

    def recommend(self, user_vec: np.ndarray, top_n: int):
        u = np.ascontiguousarray(user_vec, dtype=np.float32).reshape(self.dim)
        rows = np.empty(top_n, dtype=np.uint64)
        rec = np.empty(top_n, dtype=np.float32)
        rnk = np.empty(top_n, dtype=np.float32)
        fus = np.empty(top_n, dtype=np.float64)
        cnt = C.c_uint32()
        _lib.check(self.L.pg_router_recommend(self.h, _ptr(u), top_n, _ptr(rows), _ptr(rec), _ptr(rnk), _ptr(fus), C.byref(cnt)))
        return rows, rec, rnk, fus, cnt.value

    def recall(self, query: np.ndarray):
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.dim)
        rows = np.empty(self.k, dtype=np.uint64)
        scores = np.empty(self.k, dtype=np.float32)
        cnt = C.c_uint32()
        _lib.check(self.L.pg_router_recall(self.h, _ptr(q), _ptr(rows), _ptr(scores), C.byref(cnt)))
        return rows, scores, cnt.value

    def served(self) -> np.ndarray:
        out = (C.c_uint64 * len(self.replicas))()
        _lib.check(self.L.pg_router_stats(self.h, out))
        out = np.array(list(out), dtype=np.uint64)
        return out


class ShardGroup:
    """pg_group_*: the item table in row-range shards over several GPUs of this process (or logical shards of one)."""

    def __init__(self, devices: Sequence[int]):
        self.L = _lib.load()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _lib.check(self.L.pg_group_create(arr, len(devices), C.byref(h)))
        self.h, self.n = h, len(devices)
        self.dim = 0

    def destroy(self):
        if self.h:
            _lib.check(self.L.pg_group_destroy(self.h))
            self.h = None

    def table_create(self, total_rows: int, dim: int):
        _lib.check(self.L.pg_group_table_create(self.h, total_rows, dim))
        self.dim = dim

    def table_fill_synthetic(self, seed: int, normalize: bool = True):
        _lib.check(self.L.pg_group_table_fill_synthetic(self.h, seed, int(normalize)))

    def table_upload(self, rows: np.ndarray, row0: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        _lib.check(self.L.pg_group_table_upload(self.h, row0, rows.shape[0], _ptr(rows)))

    def model_load(self, kind: int, prec: int, blob: bytes):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _lib.check(self.L.pg_group_model_load(self.h, kind, prec, buf, len(blob)))

    def recommend(self, expr: "Expr", rank_var: str, user_vecs: np.ndarray, k: int, top_n: int,
                  dpp_candidates: int = 0, dpp_alpha: float = 1.0, dpp_window: int = 10,
                  dpp_normalize_emb: bool = True):
        u = np.ascontiguousarray(user_vecs, dtype=np.float32).reshape(-1, self.dim)
        nq = u.shape[0]
        plan = _lib.PgGroupPlan(k, dpp_candidates, dpp_alpha, dpp_window, int(dpp_normalize_emb))
        rows = np.empty((nq, top_n), dtype=np.uint64)
        rec = np.empty((nq, top_n), dtype=np.float32)
        rnk = np.empty((nq, top_n), dtype=np.float32)
        fus = np.empty((nq, top_n), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.uint32)
        _lib.check(self.L.pg_group_recommend(self.h, expr.h, rank_var.encode(), C.byref(plan), _ptr(u), nq, top_n,
                                             _ptr(rows), _ptr(rec), _ptr(rnk), _ptr(fus), _ptr(cnt)))
        return rows, rec, rnk, fus, cnt


    def recommend_begin(self, expr: "Expr", rank_var: str, user_vecs: np.ndarray, k: int, top_n: int,
                        dpp_candidates: int = 0, dpp_alpha: float = 1.0, dpp_window: int = 10,
                        dpp_normalize_emb: bool = True):
        """pg_group_recommend_begin: enqueue a step, return its ticket (up to two may be outstanding)."""
        u = np.ascontiguousarray(user_vecs, dtype=np.float32).reshape(-1, self.dim)
        plan = _lib.PgGroupPlan(k, dpp_candidates, dpp_alpha, dpp_window, int(dpp_normalize_emb))
        tk = C.c_void_p()
        _lib.check(self.L.pg_group_recommend_begin(self.h, expr.h, rank_var.encode(), C.byref(plan), _ptr(u), u.shape[0], top_n,
                                                   C.byref(tk)))
        return (tk, u.shape[0], top_n)

    def exchange_stats(self) -> dict:
        """the first exchange: steps served, steps repeated with the whole lists, bytes per shard and peer and entries per request in the last step"""
        a = (C.c_uint64 * 4)()
        _lib.check(self.L.pg_group_exchange_stats(self.h, a))
        return {"steps": int(a[0]), "round2_steps": int(a[1]), "exchange1_bytes_per_shard": int(a[2]), "entries_per_request_and_shard": int(a[3])}

    def recommend_end(self, ticket):
        tk, nq, top_n = ticket
        rows = np.empty((nq, top_n), dtype=np.uint64)
        rec = np.empty((nq, top_n), dtype=np.float32)
        rnk = np.empty((nq, top_n), dtype=np.float32)
        fus = np.empty((nq, top_n), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.uint32)
        _lib.check(self.L.pg_group_recommend_end(self.h, tk, _ptr(rows), _ptr(rec), _ptr(rnk), _ptr(fus), _ptr(cnt)))
        return rows, rec, rnk, fus, cnt


def dpp(ctx: Context, table: Table, cand_rows, rel, alpha: float, topn: int, window: int,
        normalize_emb: bool = True) -> np.ndarray:
    c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
    r = np.ascontiguousarray(rel, dtype=np.float64)
    out = np.zeros(max(topn, 1), dtype=np.uint32)
    cnt = C.c_uint32()
    _lib.check(ctx.L.pg_dpp(ctx.h, table.h, _ptr(c), _ptr(r), c.shape[0], alpha, topn, window,
                            int(normalize_emb), _ptr(out), C.byref(cnt)))
    return out[:cnt.value]


def dpp_ex(ctx: Context, table: Optional[Table], cand_rows, rel, alpha: float, topn: int, window: int,
           normalize_emb: bool = True, ensure_pos_similarity: bool = True, norm_relevance_score: int = 0,
           hook_emb: Optional[np.ndarray] = None):
    """pg_dpp_ex: DPPSort.KernelMatrix + DPPWithWindow with every option.  table=None → hook embeddings only.
    Returns (picked indices, relevance scores as used)."""
    r = np.ascontiguousarray(rel, dtype=np.float64)
    n = r.shape[0]
    c = np.ascontiguousarray(cand_rows, dtype=np.uint32) if table is not None else None
    h = None if hook_emb is None else np.ascontiguousarray(hook_emb, dtype=np.float64).reshape(n, -1)
    opt = _lib.PgDppOptions(alpha, topn, window, int(normalize_emb), int(ensure_pos_similarity),
                            int(norm_relevance_score), int(table is not None), 0 if h is None else h.shape[1])
    out = np.zeros(max(topn, 1), dtype=np.uint32)
    used = np.zeros(max(n, 1), dtype=np.float64)
    cnt = C.c_uint32()
    _lib.check(ctx.L.pg_dpp_ex(ctx.h, table.h if table is not None else None, _ptr(c) if c is not None else None,
                               _ptr(r), n, C.byref(opt), _ptr(h) if h is not None else None, _ptr(out),
                               C.byref(cnt), _ptr(used)))
    return out[:cnt.value], used[:n]


def ssd(ctx: Context, table: Table, cand_rows, rel, gamma: float, topn: int, window: int,
        normalize_emb: bool = True, ensure_pos_similarity: bool = True, norm_quality_score: int = 0,
        use_ssd_star: bool = False):
    """SSDSort.SSDWithSlidingWindow over candidates given in score-descending order.
    Returns (picked indices, quality scores)."""
    c = np.ascontiguousarray(cand_rows, dtype=np.uint32)
    r = np.ascontiguousarray(rel, dtype=np.float64)
    out = np.zeros(max(c.shape[0], 1), dtype=np.uint32)
    qual = np.zeros(max(c.shape[0], 1), dtype=np.float64)
    cnt = C.c_uint32()
    _lib.check(ctx.L.pg_ssd(ctx.h, table.h, _ptr(c), _ptr(r), c.shape[0], gamma, topn, window,
                            int(normalize_emb), int(ensure_pos_similarity), int(norm_quality_score),
                            int(use_ssd_star), _ptr(out), C.byref(cnt), _ptr(qual)))
    return out[:cnt.value], qual[:c.shape[0]]


F_I32, F_I64, F_F32, F_F64 = 1, 2, 3, 4
_F_NP = {F_I32: np.int32, F_I64: np.int64, F_F32: np.float32, F_F64: np.float64}


class Features:
    """Typed item-feature columns in HBM (pg_features_*): the device form of the reference's per-request
    "context features" (service/rank/algo_data.go:223-306)."""

    def __init__(self, ctx: Context, rows: int):
        self.ctx, self.rows = ctx, rows
        h = C.c_void_p()
        _lib.check(ctx.L.pg_features_create(ctx.h, rows, C.byref(h)))
        self.h = h

    def destroy(self):
        if self.h:
            _lib.check(self.ctx.L.pg_features_destroy(self.ctx.h, self.h))
            self.h = None

    def set_column(self, name: str, dtype: int, values: Optional[np.ndarray] = None, default: float = 0.0):
        v = None
        if values is not None:
            v = np.ascontiguousarray(values, dtype=_F_NP[dtype])
            assert v.shape == (self.rows,)
        _lib.check(self.ctx.L.pg_features_set_column(self.ctx.h, self.h, name.encode(), dtype,
                                                     _ptr(v) if v is not None else None, float(default)))

    def index(self, name: str) -> int:
        return int(self.ctx.L.pg_features_column_index(self.h, name.encode()))

    def _cols(self, names) -> np.ndarray:
        idx = np.asarray([self.index(n) if isinstance(n, str) else int(n) for n in names], dtype=np.int32)
        return idx

    def eval_expr(self, expr: "Expr", rows: np.ndarray) -> np.ndarray:
        """The expression with its variables bound to the columns of their names at `rows`, fp64 (pg_features_eval_dev):
        a numeric `expression` normalizer over item features for a candidate batch."""
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        d_r = self.ctx.to_device(r)
        d_o = self.ctx.malloc(max(r.shape[0] * 8, 16))
        try:
            _lib.check(self.ctx.L.pg_features_eval_dev(self.ctx.h, self.h, expr.h, d_r, r.shape[0], d_o))
            out = np.zeros(r.shape[0], dtype=np.float64)
            self.ctx.d2h(out, d_o)
        finally:
            self.ctx.free(d_r)
            self.ctx.free(d_o)
        return out

    def gather_i32(self, names, rows: np.ndarray) -> np.ndarray:
        idx = self._cols(names)
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        d_r = self.ctx.to_device(r)
        d_o = self.ctx.malloc(max(r.shape[0] * idx.shape[0] * 4, 16))
        _lib.check(self.ctx.L.pg_features_gather_i32_dev(self.ctx.h, self.h, _ptr(idx), idx.shape[0], d_r,
                                                         r.shape[0], d_o))
        out = np.zeros((r.shape[0], idx.shape[0]), dtype=np.int32)
        self.ctx.d2h(out, d_o)
        self.ctx.free(d_r)
        self.ctx.free(d_o)
        return out

    def gather_f32(self, names, rows: np.ndarray, scale=None, bias=None) -> np.ndarray:
        idx = self._cols(names)
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
        bi = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        d_r = self.ctx.to_device(r)
        d_o = self.ctx.malloc(max(r.shape[0] * idx.shape[0] * 4, 16))
        _lib.check(self.ctx.L.pg_features_gather_f32_dev(self.ctx.h, self.h, _ptr(idx), idx.shape[0],
                                                         _ptr(sc) if sc is not None else None,
                                                         _ptr(bi) if bi is not None else None, d_r, r.shape[0], d_o))
        out = np.zeros((r.shape[0], idx.shape[0]), dtype=np.float32)
        self.ctx.d2h(out, d_o)
        self.ctx.free(d_r)
        self.ctx.free(d_o)
        return out
