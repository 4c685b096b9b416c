"""world_size-2 test of the multi-GPU orchestration (pairec_amd/dist.py) on CPU with gloo.

The step function under test is the product's; only the engine behind it is a CPU stand-in built
on the oracle, so what is verified here is everything that is multi-rank specific: shard ranges,
the all_gather layout handed to the merge, ownership bookkeeping, request-order preservation of
the owner-computes rank, the all_reduce'd score slab, and that every rank ends with the same,
oracle-identical answer."""
import os

os.environ.setdefault("OMP_NUM_THREADS", "2")     # two ranks share this box's cores
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as o                       # noqa: E402
from pairec_amd.dist import shard_range, sharded_step  # noqa: E402

N_ROWS, DIM, K, NQ = 6001, 64, 300, 5


class CpuEngine:
    """Oracle-backed engine with the interface of GpuShardEngine (CPU tensors)."""

    def __init__(self, tab_shard, row_offset, weights):
        self.tab, self.off, self.w = tab_shard, row_offset, weights

    def recall_local(self, queries, nq, k):
        r, s = o.recall_topk(self.tab, queries.numpy(), k, row_offset=self.off)
        rows = np.full((nq, k), -1, dtype=np.int64)                 # UINT64_MAX padding
        scores = np.full((nq, k), -np.inf, dtype=np.float32)
        rows[:, :r.shape[1]] = r.astype(np.int64)
        scores[:, :s.shape[1]] = s
        return torch.from_numpy(rows), torch.from_numpy(scores)

    def merge(self, g_rows, g_scores, k):
        nq = g_rows.shape[0]
        out_r = np.zeros((nq, k), dtype=np.int64)
        out_s = np.zeros((nq, k), dtype=np.float32)
        for q in range(nq):
            rr, ss = g_rows[q].numpy().reshape(-1), g_scores[q].numpy().reshape(-1)
            keep = rr >= 0
            r, s = o.topk_merge(rr[keep].astype(np.uint64)[None], ss[keep][None], k)
            out_r[q], out_s[q] = r.astype(np.int64), s
        return torch.from_numpy(out_r), torch.from_numpy(out_s)

    def rows_to_local(self, rows):
        r = rows.numpy()
        owned = (r >= self.off) & (r < self.off + self.tab.shape[0])
        local = np.where(owned, r - self.off, 0).astype(np.int32)
        return torch.from_numpy(local), torch.from_numpy(owned)

    def rank(self, queries, local_compact, req_offsets, nq, n_items):
        ro = req_offsets.numpy()
        out = np.zeros(n_items, dtype=np.float32)
        for r in range(nq):
            a, b = int(ro[r]), int(ro[r + 1])
            if b > a:
                user = np.tile(queries[r].numpy(), 2)                # d_user = 128 from a 64-d query
                out[a:b] = o.dnn3_forward(self.w, 0, user, np.tile(self.tab[local_compact[a:b].numpy()], (1, 2)))
        return torch.from_numpy(out)

    def fuse_sort(self, rank_scores, recall_scores, nq, k):
        rs = o.widen_f32(rank_scores.numpy())
        cs = o.widen_f32(recall_scores.numpy())
        fused = rs * (1 + cs) ** 0.1
        order = np.stack([o.sort_scores(fused[q], True) for q in range(nq)]).astype(np.int32)
        return torch.from_numpy(fused), torch.from_numpy(order)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tab = o.synth_rows(o.SEED_TABLE, 0, N_ROWS, DIM)
        b, e = shard_range(N_ROWS, world, rank)
        eng = CpuEngine(tab[b:e], b, o.Dnn3Weights())
        queries = torch.from_numpy(o.synth_rows(o.SEED_QUERY, 0, NQ, DIM))
        rows, fused, order = sharded_step(eng, dist if world > 1 else None, torch, queries, NQ, K)
        q.put((rank, rows.numpy(), fused.numpy(), order.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_shard_ranges_cover_table():
    for total, world in ((100_000_000, 8), (6001, 2), (7, 3), (5, 8)):
        r = [shard_range(total, world, i) for i in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


@pytest.mark.timeout(300)
def test_sharded_step_world2_equals_single():
    single = _run(1)[0]
    two = _run(2)
    # every rank ends with the same answer, and it is the single-shard answer
    for rank, rows, fused, order in two:
        assert np.array_equal(rows, single[1]), rank
        assert np.array_equal(fused.view(np.uint64), single[2].view(np.uint64)), rank
        assert np.array_equal(order, single[3]), rank
    # and the recall part is the oracle's global top-K
    tab = o.synth_rows(o.SEED_TABLE, 0, N_ROWS, DIM)
    g_rows, _ = o.recall_topk(tab, o.synth_rows(o.SEED_QUERY, 0, NQ, DIM), K)
    assert np.array_equal(single[1].astype(np.uint64), g_rows)
