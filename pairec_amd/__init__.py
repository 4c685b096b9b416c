"""pairec_amd — MI355X-native engine for pairec's rank + recall hot path.

The product is libpairec_gpu.so (C ABI, include/pairec_gpu.h) built from pairec_amd/csrc.
This package is the Python host mirror used by tests and bench.py; see INTEGRATION.md for the
cgo shim that plugs the same library under pairec's algorithm/recall/sort registries.
"""
from . import _lib  # noqa: F401
from .engine import (Context, Table, RankModel, Expr, Features, ItemRows, Coalescer, GroupCoalescer, Router, ShardGroup, recommend_dnn3, dpp, dpp_ex, ssd, pack_dnn3, pack_dnn3_multi, pack_fm2t,  # noqa: F401
                     F_I32, F_I64, F_F32, F_F64,
                     PREC_F32, PREC_BF16, PREC_BF16X3, MODEL_DNN3, MODEL_FM_TWOTOWER, MODEL_DNN3_MULTI, MAX_QUERIES)
