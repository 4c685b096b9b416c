"""Multi-GPU orchestration of the hot path: one process per GPU, torch.distributed (backend "nccl"
= RCCL over xGMI on the GPU box, "gloo" in the CPU tests) as plumbing.

Sharding: the item table is split into contiguous row ranges, one per rank.  One
request batch then costs exactly two exchanges, both latency-bound (KB-scale), so they are single
collectives rather than anything ring-tuned:

  1. every rank scans its shard → local top-K (global row id, score) per request
  2. ONE all_gather of the HEADS of the packed lists — the best m = ceil(K/G + 6 sqrt(K/G) + 8) entries of every request
     (global rows and scores in one byte block per rank; 783 of 5 000 at G = 8: 2.4 MB per shard and 256-request step
     instead of 15.4) → identical deterministic merge on every rank → global top-K.  Exact by the same argument as the
     recall's threshold streaming: an entry a shard did NOT send ranks behind that shard's m-th entry, so when the m-th
     entry of every shard is itself outside the merged top-K nothing unsent can belong to it.  A shard whose m-th entry is
     inside (its share of the answer is > 6 sigma above K/G: rows sorted by score, one shard holding the whole answer, all
     ties) makes the step repeat the exchange with the full lists — every rank takes that decision from the same merged
     bytes; it is counted (`exchange_stats`)
  3. every rank ranks the candidates whose embedding rows it owns (no feature traffic); they are compacted on
     the device (pg_owned_compact_dev), nothing is read back
  4. all_reduce(sum) of the [R][K] score slab (each slot is written by exactly one owner, the
     other ranks contribute +0.0, so the sum is exact) → fusion + sort, replicated on every rank
  5. (cfg 5) sort.dpp_sort on the merged list, spread over the ranks BY REQUEST: the first max(page, CandidateCount)
     entries' embedding rows are contributed by their owners through a reduce_scatter (256 KB per request; rank r
     receives the summed rows of its block of requests only — half an all_reduce's traffic), every rank runs DPP on
     its R / world requests, and the picks ([R / world][page] int32) come back with an all_gather

The same flow inside ONE process over several GPUs, with direct peer stores instead of collectives, is
pg_group_recommend (csrc/group.hip) — what a cgo host calls.

The step is written against a small engine interface so that the same orchestration runs on HIP
(GpuShardEngine, the product) and, in tests/test_dist_gloo.py, on a CPU stand-in that checks the
collective layout, ownership bookkeeping and request-order preservation under world_size 2.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np


def exchange_width(k: int, world: int) -> int:
    """Entries per request a shard sends in the first exchange: its expected share of the global top-k, K/G, plus six standard
    deviations of that share for rows spread over the shards at random (binomial: sd < sqrt(K/G)) plus 8."""
    if world <= 1:
        return k
    per = k / world
    return min(k, int(np.ceil(per + 6.0 * np.sqrt(per) + 8.0)))


def tail_needed(torch, g_rows, g_scores, m_rows, m_scores, k: int):
    """[G] bool: shard g's LAST SENT entry of some request ranks strictly inside that request's merged top-k, so an entry it
    did not send could too.  Order = the recall's: score descending, row ascending; anything odd (NaN scores, padding rows in the
    last sent position or in the k-th merged one) counts as needed — the full exchange is always right."""
    s_k, r_k = m_scores[:, k - 1], m_rows[:, k - 1]                        # [nq]
    s_m, r_m = g_scores[:, :, -1], g_rows[:, :, -1]                        # [G, nq]
    inside = (s_m > s_k[None]) | ((s_m == s_k[None]) & (r_m < r_k[None]))
    odd = torch.isnan(s_m) | torch.isnan(s_k)[None] | (r_m < 0) | (r_k < 0)[None]
    return (inside | odd).any(dim=1)


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [begin, end) of `rank`; the first total_rows % world ranks get one
    extra row.  Every range starts on a multiple of 1 row; global ids must stay below 2**32."""
    base, rem = divmod(total_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_context(torch, pa, device: int):
    """A library context and the torch stream it shares, for one rank of the sharded path.

    The step interleaves torch ops (zero_, all_gather, all_reduce, gather) with library kernels; both must be
    ordered on ONE stream.  torch's default stream has handle 0, which pg_init cannot adopt (NULL means "create a
    private stream"), so the rank gets a dedicated torch.cuda.Stream: the library launches on it, and
    sharded_step runs every torch op of the step under `torch.cuda.stream(...)` of the same stream (RCCL
    collectives order themselves against the current stream)."""
    stream = torch.cuda.Stream(device=torch.device("cuda", device))
    ctx = pa.Context(device, stream.cuda_stream)
    ctx.torch_stream = stream
    return ctx, stream


class HostStagedCollectives:
    """torch.distributed's three collectives of the step with the payload staged through host memory — the DEV / TEST
    transport for several ranks that share ONE device (RCCL refuses two ranks on a device, and gloo has no
    all_gather / reduce_scatter for device tensors).  The kernels, buffers, ownership bookkeeping and the order of
    the exchanges are the product's; only the wire differs.  Never used when every rank has its own GPU.

    Every call runs on the caller's current stream: `.cpu()` waits for what the stream has queued, the copy back is
    queued behind it."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def get_world_size(self):
        return self.dist.get_world_size()

    def get_rank(self):
        return self.dist.get_rank()

    def barrier(self):
        self.dist.barrier()

    def all_gather_into_tensor(self, out, inp):
        h_out = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(h_out, inp.cpu())
        out.copy_(h_out)

    def all_reduce(self, t):
        h = t.cpu()
        self.dist.all_reduce(h)
        t.copy_(h)

    def reduce_scatter_tensor(self, out, inp):
        # (sum of disjoint owners' rows and +0.0 elsewhere: exact whatever the reduction order)
        h = inp.cpu()
        self.dist.all_reduce(h)
        n = out.shape[0]
        r = self.dist.get_rank()
        out.copy_(h[r * n:(r + 1) * n])


def sharded_step(engine, dist, torch, queries, nq: int, k: int, page: int = 0, dpp=None, prune: bool = True):
    """One request batch (see _sharded_step), with every torch op of the step on the engine's stream."""
    guard = getattr(engine, "stream_guard", None)
    if guard is None:
        return _sharded_step(engine, dist, torch, queries, nq, k, page, dpp, prune)
    with guard():
        return _sharded_step(engine, dist, torch, queries, nq, k, page, dpp, prune)


def _exchange_lists(dist, torch, world, rows, scores, nq, width):
    """all_gather of the first `width` entries of every request's list: one byte block [rows (8 B each) | scores (4 B each)]
    per rank; the gathered block of rank g is list g of the list-major [G, nq, width] layout the merge reads."""
    nb_r, nb_s = nq * width * 8, nq * width * 4
    mine = torch.empty(nb_r + nb_s, dtype=torch.uint8, device=rows.device)
    mine[:nb_r].copy_(rows[:, :width].contiguous().view(torch.uint8).reshape(-1))
    mine[nb_r:].copy_(scores[:, :width].contiguous().view(torch.uint8).reshape(-1))
    g = torch.empty(world * (nb_r + nb_s), dtype=torch.uint8, device=rows.device)
    dist.all_gather_into_tensor(g, mine)                                     # (concatenation along dim 0: every backend's layout)
    g = g.view(world, nb_r + nb_s)
    g_rows = g[:, :nb_r].contiguous().view(rows.dtype).view(world, nq, width)
    g_scores = g[:, nb_r:].contiguous().view(scores.dtype).view(world, nq, width)
    return g_rows, g_scores, nb_r + nb_s


def _sharded_step(engine, dist, torch, queries, nq: int, k: int, page: int = 0, dpp=None, prune: bool = True):
    """One request batch through recall → exchange → owner-computes rank → exchange → fuse + sort → (DPP).

    Returns (rows [nq,k] global ids, fused scores [nq,k] f64, order [nq,k] int32) — identical on every rank —
    and, with page > 0, additionally the page: positions [nq,page] into each request's list (the head of the
    sorted list, or DPPSort's picks among its first max(page, dpp["candidates"]) entries when dpp is given:
    {"candidates": C, "alpha": a, "window": w}).  `dist` is torch.distributed (or None for world_size 1).

    Nothing in the step depends on a value read back from the device: buffers are fixed-size (K-padded), the
    owned candidates are compacted on the device, so the host only enqueues (no .item(), no mask indexing)."""
    world = dist.get_world_size() if dist is not None else 1
    rows, scores = engine.recall_local(queries, nq, k)                       # [nq,k] i64 / f32
    if world > 1:
        st = getattr(engine, "exchange_stats", None)
        if st is None:
            st = engine.exchange_stats = {"steps": 0, "round2_steps": 0, "round2_shards": 0, "exchange1_bytes_per_shard": 0,
                                          "merge_entries_per_request": 0}
        m = exchange_width(k, world) if prune else k
        l_rows, l_scores = rows, scores
        g_rows, g_scores, nbytes = _exchange_lists(dist, torch, world, l_rows, l_scores, nq, m)
        rows, scores = engine.merge(g_rows, g_scores, k)
        st["steps"] += 1
        st["exchange1_bytes_per_shard"], st["merge_entries_per_request"] = nbytes, world * m
        if m < k:
            # the one value of the step the host reads back: does any shard's tail matter?  (identical on every rank: it is
            # computed from the gathered bytes)
            need = tail_needed(torch, g_rows, g_scores, rows, scores, k)
            n_need = int(need.sum().item())
            if n_need:
                st["round2_steps"] += 1
                st["round2_shards"] += n_need
                g_rows, g_scores, _ = _exchange_lists(dist, torch, world, l_rows, l_scores, nq, k)
                rows, scores = engine.merge(g_rows, g_scores, k)
    local, slot, req_offsets = engine.owned_compact(rows, nq, k)             # compacted on the device, CSR offsets
    mine = engine.rank(queries, local, req_offsets, nq, nq * k)              # n_items = upper bound
    slab = engine.scatter(mine, slot, req_offsets, nq, k)                    # [nq*k] f32, zero where not owned
    if world > 1:
        dist.all_reduce(slab)                                                # sum; owners are disjoint → exact
    fused, order = engine.fuse_sort(slab.view(nq, k), scores, nq, k)
    if page <= 0:
        return rows, fused, order
    if dpp is None:
        return rows, fused, order, order[:, :page]
    n_cand = min(k, max(page, int(dpp["candidates"])))
    c_rows, c_rel = engine.dpp_candidates(order, rows, fused, nq, k, n_cand)
    emb = engine.gather_owned(c_rows, nq * n_cand)                           # [nq*C, dim] f32, zero where not owned
    alpha, window = float(dpp["alpha"]), int(dpp["window"])
    if world == 1:
        picks = engine.dpp(emb, c_rel, nq, n_cand, alpha, page, window)      # [nq,page] into the head
        return rows, fused, order, torch.gather(order, 1, picks.long()).to(order.dtype)
    # DPP by request: rank r takes requests [r * per, (r + 1) * per).  reduce_scatter hands it the summed embedding rows of
    # exactly those (owners are disjoint, the others contribute +0.0: exact), all_gather returns everybody's picks.
    rank = dist.get_rank()
    per = (nq + world - 1) // world
    dim = emb.shape[1]
    if per * world != nq:                                                    # pad the request axis to a multiple of world
        padded = torch.zeros((per * world * n_cand, dim), dtype=emb.dtype, device=emb.device)
        padded[:nq * n_cand].copy_(emb)
        emb = padded
    my_emb = torch.empty((per * n_cand, dim), dtype=emb.dtype, device=emb.device)
    dist.reduce_scatter_tensor(my_emb, emb.contiguous())                     # (chunks along dim 0)
    q0 = rank * per
    n_mine = max(0, min(per, nq - q0))
    my_picks = torch.zeros((per, page), dtype=torch.int32, device=emb.device)
    if n_mine > 0:
        p = engine.dpp(my_emb[:n_mine * n_cand], c_rel[q0 * n_cand:(q0 + n_mine) * n_cand], n_mine, n_cand, alpha, page, window)
        my_picks[:n_mine].copy_(p.to(torch.int32))
    g_picks = torch.empty((per * world, page), dtype=torch.int32, device=emb.device)
    dist.all_gather_into_tensor(g_picks, my_picks)
    picks = g_picks[:nq]
    return rows, fused, order, torch.gather(order, 1, picks.long()).to(order.dtype)


class GpuShardEngine:
    """The product engine: every stage is a C-ABI call on HBM-resident torch tensors."""

    def __init__(self, torch, ctx, table, model, expr, k_max: int, nq_max: int):
        self.torch, self.ctx, self.table, self.model, self.expr = torch, ctx, table, model, expr
        dev = torch.device("cuda", ctx.device)
        self.dev = dev
        # the torch stream the context launches on (shard_context): torch ops and library kernels of a step are
        # ordered by being on the same stream, nothing else orders them
        self.stream = getattr(ctx, "torch_stream", None)
        if self.stream is None or self.stream.cuda_stream != ctx.stream_handle:
            raise ValueError("GpuShardEngine: the context must share a torch.cuda.Stream (use dist.shard_context); "
                             "a context with a private stream would race with the torch ops of the step")
        self.k_max = k_max
        n = nq_max * k_max
        i32, i64, f32, f64 = torch.int32, torch.int64, torch.float32, torch.float64
        self.t_rows = torch.empty((nq_max, k_max), dtype=i64, device=dev)
        self.t_scores = torch.empty((nq_max, k_max), dtype=f32, device=dev)
        self.m_rows = torch.empty((nq_max, k_max), dtype=i64, device=dev)
        self.m_scores = torch.empty((nq_max, k_max), dtype=f32, device=dev)
        self.local = torch.empty(n, dtype=i32, device=dev)
        self.slot = torch.empty(n, dtype=i32, device=dev)
        self.req_off = torch.empty(nq_max + 1, dtype=i32, device=dev)
        self.rank_out = torch.empty(n, dtype=f32, device=dev)
        self.slab = torch.empty(n, dtype=f32, device=dev)
        self.vars = torch.empty((2, n), dtype=f64, device=dev)
        self.fused = torch.empty(n, dtype=f64, device=dev)
        self.order = torch.empty(n, dtype=i32, device=dev)
        self.seg = torch.arange(0, n + 1, k_max, dtype=i32, device=dev)
        self.c_rows = self.c_rel = self.c_emb = self.picks = self.pick_cnt = None
        # (the fusion is pg_fuse_scores_dev: the RankScore's variables bind to the plane "gpu_dnn" and current_score, and a
        #  RankConfig.ScoreRewrite attached to the expression is evaluated in front of it, as in every other pipeline)
        torch.cuda.synchronize(dev)                                 # (arange above ran on torch's default stream)

    def _check(self, rc):
        from . import _lib
        _lib.check(rc)

    def stream_guard(self):
        return self.torch.cuda.stream(self.stream)

    def recall_local(self, queries, nq, k):
        rows, scores = self.t_rows[:nq, :k], self.t_scores[:nq, :k]
        assert rows.is_contiguous()
        self.table.recall_topk_dev(queries.data_ptr(), nq, k, rows.data_ptr(), scores.data_ptr())
        return rows, scores

    def merge(self, g_rows, g_scores, k):
        G, nq, per = g_rows.shape                                   # list-major, as all-gathered
        rows, scores = self.m_rows[:nq, :k], self.m_scores[:nq, :k]
        self._check(self.ctx.L.pg_topk_merge_lists_dev(self.ctx.h, g_rows.data_ptr(), g_scores.data_ptr(), nq, G,
                                                       per, 1, k, rows.data_ptr(), scores.data_ptr()))
        return rows, scores

    def owned_compact(self, rows, nq, k):
        self._check(self.ctx.L.pg_owned_compact_dev(self.ctx.h, self.table.h, rows.data_ptr(), nq, k,
                                                    self.local.data_ptr(), self.slot.data_ptr(),
                                                    self.req_off.data_ptr()))
        return self.local, self.slot, self.req_off[:nq + 1]

    def rank(self, queries, local_compact, req_offsets, nq, n_items):
        self.model.rank_dnn3_dev(self.table, queries.data_ptr(), local_compact.data_ptr(),
                                 req_offsets.data_ptr(), nq, n_items, self.rank_out.data_ptr())
        return self.rank_out

    def scatter(self, mine, slot, req_offsets, nq, k):
        slab = self.slab[:nq * k]
        slab.zero_()
        self._check(self.ctx.L.pg_scatter_f32_dev(self.ctx.h, mine.data_ptr(), slot.data_ptr(),
                                                  req_offsets[nq:].data_ptr(), nq * k, slab.data_ptr()))
        return slab

    def fuse_sort(self, rank_scores, recall_scores, nq, k):
        n = nq * k
        L, h = self.ctx.L, self.ctx.h
        fused, order = self.fused[:n], self.order[:n]
        names = (C.c_char_p * 1)(b"gpu_dnn")
        self._check(L.pg_fuse_scores_dev(h, self.expr.h, names, 1, rank_scores.contiguous().data_ptr(), n,
                                         recall_scores.contiguous().data_ptr(), n, fused.data_ptr()))
        seg = self.seg[:nq + 1] if k == self.k_max else \
            self.torch.arange(0, n + 1, k, dtype=self.torch.int32, device=self.dev)
        self._check(L.pg_sort_scores_dev(h, fused.data_ptr(), seg.data_ptr(), nq, n, k, 1, order.data_ptr()))
        return fused.view(nq, k), order.view(nq, k)

    def _dpp_buffers(self, n):
        t = self.torch
        if self.c_rows is None or self.c_rows.numel() < n:
            self.c_rows = t.empty(n, dtype=t.int64, device=self.dev)
            self.c_rel = t.empty(n, dtype=t.float64, device=self.dev)
            self.c_emb = t.empty((n, self.table.dim), dtype=t.float32, device=self.dev)

    def dpp_candidates(self, order, rows, fused, nq, k, n_cand):
        self._dpp_buffers(nq * n_cand)
        c_rows, c_rel = self.c_rows[:nq * n_cand], self.c_rel[:nq * n_cand]
        self._check(self.ctx.L.pg_dpp_candidates_dev(self.ctx.h, order.contiguous().data_ptr(), rows.contiguous().data_ptr(),
                                                     fused.contiguous().data_ptr(), nq, k, n_cand, c_rows.data_ptr(),
                                                     c_rel.data_ptr()))
        return c_rows, c_rel

    def gather_owned(self, c_rows, n):
        emb = self.c_emb[:n]
        emb.zero_()
        self._check(self.ctx.L.pg_gather_owned_rows_dev(self.ctx.h, self.table.h, c_rows.data_ptr(), n, emb.data_ptr()))
        return emb

    def dpp(self, emb, c_rel, nq, n_cand, alpha, topn, window):
        t = self.torch
        if self.picks is None or self.picks.numel() < nq * topn:
            self.picks = t.empty(nq * topn, dtype=t.int32, device=self.dev)
            self.pick_cnt = t.empty(max(nq, 256), dtype=t.int32, device=self.dev)
        picks = self.picks[:nq * topn]
        self._check(self.ctx.L.pg_dpp_batch_dev(self.ctx.h, emb.data_ptr(), c_rel.data_ptr(), nq, n_cand, self.table.dim,
                                                alpha, topn, window, 1, picks.data_ptr(), self.pick_cnt.data_ptr()))
        return picks.view(nq, topn)
