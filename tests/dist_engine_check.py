"""Child process of tests/test_gpu_group.py's dist-engine tests (torch initialises first).

World size 1 (plain `python dist_engine_check.py`): the step through GpuShardEngine on one table.
World size N (started by `python -m torch.distributed.run --nproc-per-node N`, all ranks on cuda:0): every rank
holds ITS row range of the table (row_offset != 0 on every rank but the first), the product's sharded_step runs
its exchanges — gloo rendezvous, payloads staged through the host (dist.HostStagedCollectives: RCCL refuses
two ranks on one device) — and every rank must end with the single-table oracle's page; rank 0 also checks that
all ranks hold identical rows / fused scores / order / page."""
import os
import sys

import numpy as np
import torch

torch.cuda.set_device(0)
torch.zeros(1, device="cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pairec_amd as pa                                # noqa: E402
from oracle import oracle as o                          # noqa: E402
from pairec_amd.dist import (GpuShardEngine, HostStagedCollectives, shard_context, shard_range,   # noqa: E402
                             sharded_step)
from test_gpu_group import EXPR, oracle_pipeline       # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
prec = pa.PREC_BF16 if os.environ.get("PG_CHECK_PREC") == "bf16" else pa.PREC_F32
n, d, k, R, top_n, dpp_c = 80_003, 128, 300, 7, 25, 90
coll = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
    coll = HostStagedCollectives(dist, torch)
b, e = shard_range(n, world, rank)
ctx, _stream = shard_context(torch, pa, 0)
t = pa.Table(ctx, e - b, d, row_offset=b)
tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
if rank % 2 == 0:
    t.fill_synthetic(o.SEED_TABLE)                      # generated on the device from the GLOBAL row index
else:
    t.upload(tab[b:e])
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr(EXPR)
eng = GpuShardEngine(torch, ctx, t, m, ex, k, R)
for step, (user0, nq) in enumerate(((77, R), (500, R - 2))):      # two steps: the buffers are reused
    q = o.synth_rows(o.SEED_QUERY, user0, nq, d)
    tq = torch.from_numpy(q).to("cuda:0")
    torch.cuda.synchronize()                            # the upload ran on the default stream
    rows, fused, order, page = sharded_step(eng, coll, torch, tq, nq, k, top_n,
                                            {"candidates": dpp_c, "alpha": 1.0, "window": 10})
    torch.cuda.synchronize()
    rows_h, fused_h = rows.cpu().contiguous(), fused.cpu().contiguous()
    order_h, page_h = order.cpu().contiguous().to(torch.int64), page.cpu().contiguous().to(torch.int64)
    rows_n, page_n = rows_h.numpy().astype(np.uint64), page_h.numpy()
    if prec == pa.PREC_F32:
        want = oracle_pipeline(tab, w, prec, q, k, top_n, dpp_c, 1.0, 10)
        for r in range(nq):
            assert np.array_equal(rows_n[r][page_n[r]], want[r][0]), "rank %d step %d request %d: page differs" % (rank, step, r)
    else:
        orow, _ = o.recall_topk(tab, q, k)
        assert np.array_equal(rows_n, orow), "rank %d: merged recall differs from the oracle's global top-K" % rank
    if world > 1:
        for name, x in (("rows", rows_h), ("fused", fused_h.view(torch.int64)), ("order", order_h), ("page", page_h)):
            g = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(g, x)
            if rank == 0:
                for r_, y in enumerate(g):
                    assert torch.equal(y, x), "step %d: %s of rank %d differs from rank 0's" % (step, name, r_)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
if rank == 0:
    print("dist engine OK (world %d)" % world)
