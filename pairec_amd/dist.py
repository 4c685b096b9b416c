"""Multi-GPU orchestration of the hot path: one process per GPU, torch.distributed (backend "nccl"
= RCCL over xGMI on the GPU box, "gloo" in the CPU tests) as plumbing.

Sharding: the item table is split into contiguous row ranges, one per rank.  One
request batch then costs exactly two exchanges, both latency-bound (KB-scale), so they are single
collectives rather than anything ring-tuned:

  1. every rank scans its shard → local top-K (global row id, score) per request
  2. all_gather of the [R][K] lists → identical deterministic merge on every rank → global top-K
  3. every rank ranks the candidates whose embedding rows it owns (no feature traffic)
  4. all_reduce(sum) of the [R][K] score slab (each slot is written by exactly one owner, the
     other ranks contribute +0.0, so the sum is exact) → fusion + sort, replicated on every rank

The step is written against a small engine interface so that the same orchestration runs on HIP
(GpuShardEngine, the product) and, in tests/test_dist_gloo.py, on a CPU stand-in that checks the
collective layout, ownership bookkeeping and request-order preservation under world_size 2.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [begin, end) of `rank`; the first total_rows % world ranks get one
    extra row.  Every range starts on a multiple of 1 row; global ids must stay below 2**32."""
    base, rem = divmod(total_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def sharded_step(engine, dist, torch, queries, nq: int, k: int):
    """One request batch through recall → exchange → rank → exchange → fuse+sort.

    Returns (rows [nq,k] global ids, fused scores [nq,k] f64, order [nq,k] int32) — identical on
    every rank.  `dist` is torch.distributed (or None for world_size 1)."""
    world = dist.get_world_size() if dist is not None else 1
    rows, scores = engine.recall_local(queries, nq, k)                       # [nq,k] i64 / f32
    if world > 1:
        # concatenation along dim 0 (the layout every backend accepts), viewed as [G, nq, k]
        g_rows = torch.empty((world * nq, k), dtype=rows.dtype, device=rows.device)
        g_scores = torch.empty((world * nq, k), dtype=scores.dtype, device=scores.device)
        dist.all_gather_into_tensor(g_rows, rows.contiguous())
        dist.all_gather_into_tensor(g_scores, scores.contiguous())
        # [G,nq,k] → [nq,G,k]: the merge wants all of a request's lists contiguous
        rows, scores = engine.merge(g_rows.view(world, nq, k).permute(1, 0, 2).contiguous(),
                                    g_scores.view(world, nq, k).permute(1, 0, 2).contiguous(), k)
    local, owned = engine.rows_to_local(rows)                                # [nq,k] i32 / bool
    counts = owned.sum(dim=1, dtype=torch.int32)
    req_offsets = torch.zeros(nq + 1, dtype=torch.int32, device=rows.device)
    req_offsets[1:] = torch.cumsum(counts, 0)
    n_items = int(req_offsets[-1].item())
    slab = torch.zeros(nq * k, dtype=torch.float32, device=rows.device)
    if n_items:
        mine = engine.rank(queries, local[owned].contiguous(), req_offsets, nq, n_items)
        slab[owned.reshape(-1)] = mine                                       # request order kept
    if world > 1:
        dist.all_reduce(slab)                                                # sum; owners are disjoint
    fused, order = engine.fuse_sort(slab.view(nq, k), scores, nq, k)
    return rows, fused, order


class GpuShardEngine:
    """The product engine: every stage is a C-ABI call on HBM-resident torch tensors."""

    def __init__(self, torch, ctx, table, model, expr, k_max: int, nq_max: int):
        self.torch, self.ctx, self.table, self.model, self.expr = torch, ctx, table, model, expr
        dev = torch.device("cuda", ctx.device)
        self.dev = dev
        self.k_max = k_max
        n = nq_max * k_max
        self.t_rows = torch.empty((nq_max, k_max), dtype=torch.int64, device=dev)
        self.t_scores = torch.empty((nq_max, k_max), dtype=torch.float32, device=dev)
        self.m_rows = torch.empty((nq_max, k_max), dtype=torch.int64, device=dev)
        self.m_scores = torch.empty((nq_max, k_max), dtype=torch.float32, device=dev)
        self.local = torch.empty((nq_max, k_max), dtype=torch.int32, device=dev)
        self.owned = torch.empty((nq_max, k_max), dtype=torch.uint8, device=dev)
        self.rank_out = torch.empty(n, dtype=torch.float32, device=dev)
        self.vars = torch.empty((2, n), dtype=torch.float64, device=dev)
        self.fused = torch.empty(n, dtype=torch.float64, device=dev)
        self.order = torch.empty(n, dtype=torch.int32, device=dev)
        self.seg = torch.arange(0, n + 1, k_max, dtype=torch.int32, device=dev)
        assert expr.var_names == ["gpu_dnn", "current_score"]       # column order of the vars slab below

    def _check(self, rc):
        from . import _lib
        _lib.check(rc)

    def recall_local(self, queries, nq, k):
        rows, scores = self.t_rows[:nq, :k], self.t_scores[:nq, :k]
        assert rows.is_contiguous()
        self.table.recall_topk_dev(queries.data_ptr(), nq, k, rows.data_ptr(), scores.data_ptr())
        return rows, scores

    def merge(self, g_rows, g_scores, k):
        nq, G, per = g_rows.shape
        rows, scores = self.m_rows[:nq, :k], self.m_scores[:nq, :k]
        self._check(self.ctx.L.pg_topk_merge_dev(self.ctx.h, g_rows.data_ptr(), g_scores.data_ptr(), nq, G,
                                                 per, k, rows.data_ptr(), scores.data_ptr()))
        return rows, scores

    def rows_to_local(self, rows):
        nq, k = rows.shape
        local, owned = self.local[:nq, :k], self.owned[:nq, :k]
        self._check(self.ctx.L.pg_rows_to_local_dev(self.ctx.h, self.table.h, rows.data_ptr(), nq * k,
                                                    local.data_ptr(), owned.data_ptr()))
        return local, owned.bool()

    def rank(self, queries, local_compact, req_offsets, nq, n_items):
        out = self.rank_out[:n_items]
        self.model.rank_dnn3_dev(self.table, queries.data_ptr(), local_compact.data_ptr(),
                                 req_offsets.data_ptr(), nq, n_items, out.data_ptr())
        return out

    def fuse_sort(self, rank_scores, recall_scores, nq, k):
        n = nq * k
        L, h = self.ctx.L, self.ctx.h
        v = self.vars[:, :n] if n == self.vars.shape[1] else self.torch.empty((2, n), dtype=self.torch.float64,
                                                                            device=self.dev)
        self._check(L.pg_widen_f32_dev(h, rank_scores.contiguous().data_ptr(), n, v[0].data_ptr()))
        self._check(L.pg_widen_f32_dev(h, recall_scores.contiguous().data_ptr(), n, v[1].data_ptr()))
        fused, order = self.fused[:n], self.order[:n]
        self._check(L.pg_expr_eval_dev(h, self.expr.h, v.data_ptr(), n, fused.data_ptr()))
        seg = self.seg[:nq + 1] if k == self.k_max else \
            self.torch.arange(0, n + 1, k, dtype=self.torch.int32, device=self.dev)
        self._check(L.pg_sort_scores_dev(h, fused.data_ptr(), seg.data_ptr(), nq, n, k, 1, order.data_ptr()))
        return fused.view(nq, k), order.view(nq, k)
