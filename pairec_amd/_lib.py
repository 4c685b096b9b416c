"""ctypes binding of libpairec_gpu.so (include/pairec_gpu.h).

The library is the product: there is no CPU fallback.  Importing this module without the built
extension, or calling it without a gfx950 GPU, fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PG_LIB_VARIANT=dev selects libpairec_gpu_dev.so: what `make WS_EXTRA=… / SCAN_EXTRA=… / MLP_EXTRA=…` builds (profile marks,
# ablations) — the product library is never one of those (pairec_amd/csrc/Makefile)
LIB_PATH = os.path.join(_HERE, "libpairec_gpu_dev.so" if os.environ.get("PG_LIB_VARIANT") == "dev" else "libpairec_gpu.so")
if os.environ.get("PG_LIB_PATH"):                  # developer A/B runs: an explicitly named build
    LIB_PATH = os.environ["PG_LIB_PATH"]

# every symbol include/pairec_gpu.h declares (tests/test_abi.py checks the two stay in sync)
EXPORTS = [
    "pg_last_error", "pg_version", "pg_device_count", "pg_init", "pg_shutdown", "pg_synchronize", "pg_device_malloc",
    "pg_device_free", "pg_memcpy_h2d", "pg_memcpy_d2h", "pg_table_create", "pg_table_destroy",
    "pg_table_fill_synthetic", "pg_table_upload", "pg_table_download", "pg_table_swap",
    "pg_table_info", "pg_table_gather", "pg_recall_topk", "pg_recall_topk_dev", "pg_recall_topk_l2", "pg_recall_topk_l2_dev", "pg_recall_topk_where", "pg_table_view_create",
    "pg_topk_merge_dev", "pg_model_load", "pg_model_destroy", "pg_model_num_outputs", "pg_rank_dnn3", "pg_rank_dnn3_dev",
    "pg_rank_fm2t", "pg_rank_fm2t_dev", "pg_expr_compile", "pg_expr_free", "pg_expr_num_vars", "pg_expr_set_score_rewrites", "pg_expr_compile_typed", "pg_expr_is_antlr", "pg_fuse_scores_dev",
    "pg_expr_var_name", "pg_expr_eval", "pg_expr_eval_dev", "pg_sort_scores", "pg_sort_scores_dev",
    "pg_dpp", "pg_stats", "pg_last_scan_kernel_ms", "pg_rows_to_local_dev", "pg_widen_f32_dev",
    "pg_hbm_read_probe", "pg_table_screen_info", "pg_ssd", "pg_features_create", "pg_features_destroy", "pg_features_set_column",
    "pg_features_column_index", "pg_features_num_columns", "pg_features_gather_i32_dev",
    "pg_features_gather_f32_dev", "pg_rank_fm2t_rows_dev", "pg_rank_fm2t_rows", "pg_recommend_dnn3_dev", "pg_set_option",
    "pg_table_fill_gaussian", "pg_table_fill_mixture", "pg_group_exchange_stats", "pg_dpp_ex", "pg_i2i_recall", "pg_online_vector_recall", "pg_fm2t_user_embedding",
    "pg_fm2t_user_embedding_dev", "pg_recommend_dnn3_begin", "pg_recommend_end",
    "pg_coalescer_create", "pg_coalescer_destroy", "pg_coalescer_recall", "pg_coalescer_rank_dnn3",
    "pg_coalescer_recommend", "pg_coalescer_stats", "pg_coalescer_create_scene", "pg_coalescer_i2i_recall", "pg_coalescer_recall_l2",
    "pg_coalescer_online_recall", "pg_coalescer_rank", "pg_coalescer_rank_fm2t", "pg_coalescer_recommend_ex",
    "pg_coalescer_dpp", "pg_coalescer_ssd", "pg_recommend_end_timed", "pg_debug_stall",
    "pg_fm2t_item_rows_build", "pg_fm2t_item_rows_update", "pg_fm2t_item_rows_destroy", "pg_rank_fm2t_irows_dev",
    "pg_rank_fm2t_irows",
    "pg_topk_merge_lists_dev", "pg_owned_compact_dev", "pg_scatter_f32_dev", "pg_dpp_candidates_dev",
    "pg_gather_owned_rows_dev", "pg_dpp_batch_dev", "pg_dpp_kernel_matrix_dev", "pg_features_eval_dev",
    "pg_group_create", "pg_group_destroy", "pg_group_size", "pg_group_ctx", "pg_group_table", "pg_group_table_create",
    "pg_group_table_fill_synthetic", "pg_group_table_upload", "pg_group_model_load", "pg_group_recommend",
    "pg_group_recommend_begin", "pg_group_recommend_end", "pg_group_info", "pg_coalescer_create_group",
    "pg_router_create", "pg_router_destroy", "pg_router_recommend", "pg_router_recall", "pg_router_stats",
]


class PgStats(C.Structure):
    _fields_ = [("recall_calls", C.c_uint64), ("recall_rows_scanned", C.c_uint64),
                ("recall_rescans", C.c_uint64), ("rank_calls", C.c_uint64),
                ("rank_items", C.c_uint64), ("sort_calls", C.c_uint64), ("sort_items", C.c_uint64),
                ("last_recall_ms", C.c_double), ("last_rank_ms", C.c_double),
                ("last_sort_ms", C.c_double), ("recall_predicted", C.c_uint64),
                ("recall_suspects", C.c_uint64), ("recall_suspect_queries", C.c_uint64), ("recall_i4m_pairs", C.c_uint64),
                ("recall_screen_overflows", C.c_uint64), ("recall_record_growths", C.c_uint64),
                ("recall_rescored", C.c_uint64), ("sort_split_calls", C.c_uint64)]


class PgDppOptions(C.Structure):
    _fields_ = [("alpha", C.c_double), ("topn", C.c_uint32), ("window", C.c_uint32), ("normalize_emb", C.c_int),
                ("ensure_pos_similarity", C.c_int), ("norm_relevance_score", C.c_int), ("has_table", C.c_int),
                ("hook_dim", C.c_uint32)]


class PgGroupPlan(C.Structure):
    _fields_ = [("k", C.c_uint32), ("dpp_candidates", C.c_uint32), ("dpp_alpha", C.c_double),
                ("dpp_window", C.c_uint32), ("dpp_normalize_emb", C.c_int)]


class PgCoalescerConfig(C.Structure):
    _fields_ = [("k", C.c_uint32), ("max_batch", C.c_uint32), ("max_wait_us", C.c_uint32), ("depth", C.c_uint32),
                ("max_top_n", C.c_uint32), ("max_rank_items", C.c_uint32), ("timeout_us", C.c_uint32)]


class PgRankAlgo(C.Structure):
    _fields_ = [("model", C.c_void_p), ("name", C.c_char_p), ("features", C.c_void_p),
                ("item_field_cols", C.POINTER(C.c_int32)), ("item_rows", C.c_void_p),
                ("output_names", C.POINTER(C.c_char_p))]


class PgSceneConfig(C.Structure):
    _fields_ = [("base", PgCoalescerConfig), ("algos", C.POINTER(PgRankAlgo)), ("n_algos", C.c_uint32),
                ("rank_score", C.c_void_p), ("rerank", C.c_int), ("rerank_candidates", C.c_uint32),
                ("dpp", PgDppOptions), ("query_model", C.c_void_p), ("trigger_table", C.c_void_p),
                ("max_rerank_items", C.c_uint32), ("max_hook_dim", C.c_uint32)]


class PgCoalescerStats(C.Structure):
    # flavours: 0 recall (vector / i2i / online), 1 rank, 2 recommend, 3 dpp
    _fields_ = [("requests", C.c_uint64 * 6), ("batches", C.c_uint64 * 6), ("largest_batch", C.c_uint64 * 6),
                ("replans", C.c_uint64), ("timeouts", C.c_uint64), ("device_ms", C.c_double * 6)]


_lib = None


def load():
    """Load libpairec_gpu.so; raise with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pairec_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
    P = C.POINTER
    L.pg_last_error.restype = C.c_char_p
    L.pg_version.restype = C.c_char_p
    sig = {
        "pg_device_count": [P(i32)],
        "pg_model_num_outputs": [vp, P(u32)],
        "pg_init": [i32, vp, P(vp)],
        "pg_shutdown": [vp],
        "pg_synchronize": [vp],
        "pg_device_malloc": [vp, sz, P(vp)],
        "pg_device_free": [vp, vp],
        "pg_memcpy_h2d": [vp, vp, vp, sz],
        "pg_memcpy_d2h": [vp, vp, vp, sz],
        "pg_table_create": [vp, u64, u32, u64, P(vp)],
        "pg_table_destroy": [vp, vp],
        "pg_table_fill_synthetic": [vp, vp, u64, i32],
        "pg_table_upload": [vp, vp, u64, u64, vp],
        "pg_table_download": [vp, vp, u64, u64, vp],
        "pg_table_swap": [vp, vp, vp],
        "pg_table_info": [vp, P(u64), P(u32), P(u64)],
        "pg_table_gather": [vp, vp, vp, u32, vp],
        "pg_recall_topk": [vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_recall_topk_dev": [vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_recall_topk_l2": [vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_recall_topk_l2_dev": [vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_recall_topk_where": [vp, vp, vp, i32, i32, C.c_longlong, i32, vp, u32, u32, vp, vp, vp],
        "pg_table_view_create": [vp, vp, vp, i32, i32, C.c_longlong, vp],
        "pg_topk_merge_dev": [vp, vp, vp, u32, u32, u32, u32, vp, vp],
        "pg_model_load": [vp, i32, i32, vp, sz, P(vp)],
        "pg_model_destroy": [vp, vp],
        "pg_rank_dnn3": [vp, vp, vp, vp, vp, vp, u32, vp],
        "pg_rank_dnn3_dev": [vp, vp, vp, vp, vp, vp, u32, u32, vp],
        "pg_rank_fm2t": [vp, vp, vp, vp, vp, vp, u32, vp],
        "pg_rank_fm2t_dev": [vp, vp, vp, vp, vp, vp, u32, u32, vp],
        "pg_expr_compile": [C.c_char_p, P(vp)],
        "pg_expr_free": [vp],
        "pg_expr_num_vars": [vp],
        "pg_expr_set_score_rewrites": [vp, u32, vp, vp],
        "pg_expr_compile_typed": [C.c_char_p, C.c_char_p, P(vp)],
        "pg_expr_is_antlr": [vp],
        "pg_fuse_scores_dev": [vp, vp, vp, u32, vp, sz, vp, u32, vp],
        "pg_expr_eval": [vp, vp, vp, u32, vp],
        "pg_expr_eval_dev": [vp, vp, vp, u32, vp],
        "pg_sort_scores": [vp, vp, vp, u32, i32, vp],
        "pg_sort_scores_dev": [vp, vp, vp, u32, u32, u32, i32, vp],
        "pg_dpp": [vp, vp, vp, vp, u32, C.c_double, u32, u32, i32, vp, vp],
        "pg_dpp_ex": [vp, vp, vp, vp, u32, P(PgDppOptions), vp, vp, vp, vp],
        "pg_ssd": [vp, vp, vp, vp, u32, C.c_double, u32, u32, i32, i32, i32, i32, vp, vp, vp],
        "pg_features_create": [vp, u64, P(vp)],
        "pg_features_destroy": [vp, vp],
        "pg_features_set_column": [vp, vp, C.c_char_p, i32, vp, C.c_double],
        "pg_features_column_index": [vp, C.c_char_p],
        "pg_features_num_columns": [vp],
        "pg_features_gather_i32_dev": [vp, vp, vp, u32, vp, u32, vp],
        "pg_features_gather_f32_dev": [vp, vp, vp, u32, vp, vp, vp, u32, vp],
        "pg_rank_fm2t_rows_dev": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp],
        "pg_rank_fm2t_rows": [vp, vp, vp, vp, vp, vp, vp, vp, u32, vp],
        "pg_recommend_dnn3_dev": [vp, vp, vp, vp, C.c_char_p, vp, u32, u32, vp, vp, vp, vp, vp, vp],
        "pg_set_option": [vp, C.c_char_p, C.c_char_p],
        "pg_table_fill_gaussian": [vp, vp, u64, C.c_float],
        "pg_table_fill_mixture": [vp, vp, u64, C.c_uint32, C.c_float],
        "pg_i2i_recall": [vp, vp, vp, u32, vp, u32, vp, vp, vp],
        "pg_online_vector_recall": [vp, vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_fm2t_user_embedding": [vp, vp, vp, u32, vp],
        "pg_fm2t_user_embedding_dev": [vp, vp, vp, u32, vp],
        "pg_recommend_dnn3_begin": [vp, vp, vp, vp, C.c_char_p, vp, u32, u32, vp, vp, vp, vp, vp, vp, P(vp)],
        "pg_recommend_end": [vp, vp, P(C.c_double)],
        "pg_coalescer_create": [vp, vp, vp, vp, C.c_char_p, P(PgCoalescerConfig), P(vp)],
        "pg_coalescer_destroy": [vp],
        "pg_coalescer_recall": [vp, vp, vp, vp, P(u32)],
        "pg_coalescer_rank_dnn3": [vp, vp, vp, u32, vp],
        "pg_coalescer_recommend": [vp, vp, u32, vp, vp, vp, vp, P(u32)],
        "pg_coalescer_stats": [vp, P(PgCoalescerStats)],
        "pg_coalescer_create_scene": [vp, vp, P(PgSceneConfig), P(vp)],
        "pg_coalescer_i2i_recall": [vp, u32, vp, vp, P(u32)],
        "pg_coalescer_recall_l2": [vp, vp, vp, vp, P(u32)],
        "pg_coalescer_online_recall": [vp, vp, vp, vp, P(u32)],
        "pg_coalescer_rank": [vp, u32, vp, vp, vp, u32, vp],
        "pg_coalescer_rank_fm2t": [vp, vp, vp, vp, u32, vp],
        "pg_coalescer_recommend_ex": [vp, vp, vp, u32, vp, vp, vp, vp, P(u32)],
        "pg_coalescer_dpp": [vp, vp, vp, u32, P(PgDppOptions), vp, vp, P(u32), vp],
        "pg_coalescer_ssd": [vp, vp, vp, u32, C.c_double, u32, u32, i32, i32, i32, i32, vp, P(u32), vp],
        "pg_recommend_end_timed": [vp, vp, u32, P(C.c_double)],
        "pg_debug_stall": [vp, u32],
        "pg_fm2t_item_rows_build": [vp, vp, vp, vp, P(vp)],
        "pg_fm2t_item_rows_update": [vp, vp, u64, u64],
        "pg_fm2t_item_rows_destroy": [vp, vp],
        "pg_rank_fm2t_irows_dev": [vp, vp, vp, vp, vp, vp, vp, u32, u32, vp],
        "pg_rank_fm2t_irows": [vp, vp, vp, vp, vp, vp, vp, u32, vp],
        "pg_topk_merge_lists_dev": [vp, vp, vp, u32, u32, u32, i32, u32, vp, vp],
        "pg_owned_compact_dev": [vp, vp, vp, u32, u32, vp, vp, vp],
        "pg_scatter_f32_dev": [vp, vp, vp, vp, u32, vp],
        "pg_dpp_candidates_dev": [vp, vp, vp, vp, u32, u32, u32, vp, vp],
        "pg_gather_owned_rows_dev": [vp, vp, vp, u32, vp],
        "pg_dpp_batch_dev": [vp, vp, vp, u32, u32, u32, C.c_double, u32, u32, i32, vp, vp],
        "pg_dpp_kernel_matrix_dev": [vp, vp, vp, u32, u32, u32, C.c_double, i32, vp],
        "pg_features_eval_dev": [vp, vp, vp, vp, u32, vp],
        "pg_group_create": [P(C.c_int), u32, P(vp)],
        "pg_group_destroy": [vp],
        "pg_group_table_create": [vp, u64, u32],
        "pg_group_table_fill_synthetic": [vp, u64, i32],
        "pg_group_table_upload": [vp, u64, u64, vp],
        "pg_group_model_load": [vp, i32, i32, vp, sz],
        "pg_group_recommend": [vp, vp, C.c_char_p, P(PgGroupPlan), vp, u32, u32, vp, vp, vp, vp, vp],
        "pg_group_recommend_begin": [vp, vp, C.c_char_p, P(PgGroupPlan), vp, u32, u32, P(vp)],
        "pg_group_recommend_end": [vp, vp, vp, vp, vp, vp, vp],
        "pg_group_info": [vp, P(u64), P(u32)],
        "pg_group_exchange_stats": [vp, P(u64)],
        "pg_coalescer_create_group": [vp, vp, C.c_char_p, P(PgGroupPlan), P(PgCoalescerConfig), P(vp)],
        "pg_router_create": [P(vp), u32, P(vp)],
        "pg_router_destroy": [vp],
        "pg_router_recommend": [vp, vp, u32, vp, vp, vp, vp, P(u32)],
        "pg_router_recall": [vp, vp, vp, vp, P(u32)],
        "pg_router_stats": [vp, P(u64)],
        "pg_rows_to_local_dev": [vp, vp, vp, u32, vp, vp],
        "pg_widen_f32_dev": [vp, vp, u32, vp],
        "pg_stats": [vp, P(PgStats)],
        "pg_last_scan_kernel_ms": [vp, P(C.c_double), P(u64)],
        "pg_hbm_read_probe": [vp, vp, i32, P(C.c_double)],
        "pg_table_screen_info": [vp, vp, P(C.c_int), P(C.c_float), P(C.c_float)],
    }
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:
        raise ImportError("pairec_amd: %s lacks symbols %s declared in include/pairec_gpu.h "
                          "(stale build?)" % (LIB_PATH, missing))
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.pg_group_size.argtypes = [vp]
    L.pg_group_size.restype = u32
    L.pg_group_ctx.argtypes = [vp, u32]
    L.pg_group_ctx.restype = vp
    L.pg_group_table.argtypes = [vp, u32]
    L.pg_group_table.restype = vp
    L.pg_expr_var_name.argtypes = [vp, i32]
    L.pg_expr_var_name.restype = C.c_char_p
    _lib = L
    return L


class PgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("pairec_gpu error %d: %s" % (code, msg))
        self.code = code


def check(rc: int):
    if rc != 0:
        raise PgError(rc, load().pg_last_error().decode("utf-8", "replace"))
