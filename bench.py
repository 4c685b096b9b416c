#!/usr/bin/env python3
"""bench.py — ranked items/sec of pairec's recall → rank → sort hot path on MI355X.

Workload (BASELINE.json configs[2], the config the metric is quoted on): a batch of R=256 requests;
each request = exact inner-product recall of the top 5000 of a 100M x 128 fp32 item table resident
in HBM → 3-layer DNN rank (256→512→256→1, bf16 MFMA) of those 5000 candidates → RankScore fusion
in fp64 → ItemRankScore (descending) sort.  A "step" is one such batch; value = ranked items / s
(R*5000 per step), inputs resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W
N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL), two modes:
  --mode replica (default): the 51.2 GB table fits one GPU, so it is replicated and the *requests*
      are sharded — every rank runs its own batches, no data-path collective, weak scaling.
  --mode shard: cfg-5 style — every rank holds --rows rows of one N x --rows table (contiguous row
      ranges), all_gather of the per-shard top-K lists + all_reduce of the score slab per step
      (pairec_amd/dist.py); value counts each request once.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 (v_mfma_i32_32x32x32_i8 issues at the bf16 rate with twice the k)
FLOPS_PER_ITEM = 524800        # SURVEY.md §8(d) cfg 3: 2*(256*512+512*256+256)
RANK_EXPR = "${gpu_dnn}*(1+${current_score})^0.1"      # RankConf.RankScore: model score x recall score


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=256, help="requests per step (<= 256 = one table pass)")
    ap.add_argument("--prec", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--mode", choices=["replica", "shard"], default="replica")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-reqs", type=int, default=200,
                    help="single-request latency samples at N=1 (the first 10 % are discarded as warm-up, SURVEY.md 8d)")
    return ap.parse_args()


def pmc_traffic(R):
    """HBM bytes per table pass of the dominant kernel from the committed rocprofv3 PMC pass of this
    same command (profiles/r1_scan_traffic.json; FETCH_SIZE x2 per the gfx950 correction of
    MI355X_MICROARCH.md §HBM, + WRITE_SIZE).  None if that profile does not cover this batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_scan_traffic.json")) as f:
            d = json.load(f)
        return d.get(str(R), {}).get("hbm_bytes_per_pass")
    except Exception:
        return None


def scan_kernel_name(R, dim, elem_bytes):
    """The kernel recall.hip dispatches for the full-table pass at this batch size (dispatch_screen):
    screen_kernel<DIM, NQB, WAVES, SPLIT, VAR, I8, QH>."""
    if elem_bytes == 0:
        return "pg::scan_kernel<%d,1>" % dim
    if elem_bytes == 1:
        if R > 128:
            return "pg::screen_kernel<128,4,8,1,0,true,2>"
        nqb = 4 if R > 64 else (2 if R > 32 else 1)
        return "pg::screen_kernel<128,%d,8,1,0,true,1>" % nqb
    nqb, waves = (8, 4) if R > 128 else ((4, 8) if R > 64 else ((2, 8) if R > 32 else (1, 8)))
    return "pg::screen_kernel<%d,%d,%d,1,0,false,1>" % (dim, nqb, waves)


def device_info():
    """Marketing name / CU count / max clock from rocminfo (child process; evidence for SURVEY.md 8(d))."""
    try:
        import subprocess
        txt = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=30).stdout
        blocks = [b for b in txt.split("Agent ") if "gfx" in b and "Device Type:             GPU" in b]
        b = blocks[0]
        def field(k):
            for line in b.splitlines():
                if line.strip().startswith(k):
                    return line.split(":", 1)[1].strip()
            return None
        return {"name": field("Marketing Name"), "arch": field("Name"), "compute_units": field("Compute Unit"),
                "max_clock_mhz": field("Max Clock Freq")}
    except Exception:
        return None


def make_queries(o, step, R, dim):
    # 1000 distinct users cycle through the run
    return o.synth_rows(o.SEED_QUERY, (step * R) % 1000, R, dim)


# ------------------------------------------------------------------------------------------------
# single-GPU pipeline on raw device pointers (no torch on this path)
# ------------------------------------------------------------------------------------------------
class Pipeline1:
    def __init__(self, pa, ctx, table, model, expr, R, K):
        self.pa, self.ctx, self.table, self.model, self.expr, self.R, self.K = pa, ctx, table, model, expr, R, K
        n = R * K
        m = ctx.malloc
        self.d_rows, self.d_scores = m(n * 8), m(n * 4)
        self.d_rank = m(n * 4)
        self.d_fused, self.d_order = m(n * 8), m(n * 4)

    def step(self, d_q, R=None):
        """One request batch = ONE call into the library (pg_recommend_dnn3_dev): recall top-K → DNN3 rank of
        every candidate → RankScore fusion → ItemRankScore sort, all device-resident."""
        from pairec_amd import _lib
        R = R or self.R
        ctx = self.ctx
        _lib.check(ctx.L.pg_recommend_dnn3_dev(ctx.h, self.table.h, self.model.h, self.expr.h, b"gpu_dnn", d_q, R,
                                               self.K, self.d_rows, self.d_scores, self.d_rank, self.d_fused,
                                               self.d_order, None))


def cpu_baseline(o, args, R, K):
    """The oracle (a C port of the reference-shaped CPU path) on a bounded sample of the same
    workload, all host cores: recall scan of a table slice (scaled to the full table by row count)
    + DNN rank of a candidate sample (scaled to R*K items) + sort.  kind = "port": the reference is
    Go with its arithmetic in remote services; nothing of it can run here."""
    cores = os.cpu_count() or 1
    slice_rows = 8_000_000 if cores >= 32 else 1_000_000
    tab = o.synth_rows(o.SEED_TABLE, 0, slice_rows, args.dim)
    q = make_queries(o, 0, R, args.dim)
    t0 = time.time()
    rows, scores = o.recall_topk(tab, q, K, threads=cores)
    t_recall_slice = time.time() - t0
    t_recall = t_recall_slice * (args.rows / slice_rows)
    w = o.Dnn3Weights()
    n_sample = 20000
    cand = tab[rows[0][:K].astype(np.int64) % slice_rows]
    items = np.tile(cand, (n_sample // K + 1, 1))[:n_sample]
    t0 = time.time()
    sc = o.dnn3_forward(w, 1 if args.prec == "bf16" else 0, q[0], items, threads=cores)
    t_rank = (time.time() - t0) * (R * K / n_sample)
    t0 = time.time()
    for r in range(R):
        o.sort_scores(sc[:K].astype(np.float64), True)
    t_sort = time.time() - t0
    total = t_recall + t_rank + t_sort
    return {
        "value": R * K / total, "unit": "ranked items/s", "cores": cores, "kind": "port",
        "sample": "recall: %d-row slice x %d queries (%.2f s) scaled x%d by rows; rank: %d items "
                  "(scaled to %d); sort: %d x %d" % (slice_rows, R, t_recall_slice, args.rows // slice_rows,
                                                     n_sample, R * K, R, K),
        "stage_seconds_per_step": {"recall": t_recall, "rank": t_rank, "sort": t_sort},
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    R, K = args.batch, args.k
    assert 1 <= R <= 256

    import pairec_amd as pa
    from oracle import oracle as o       # synthetic-data spec + cpu_baseline leg only

    torch = dist = None
    stream = None
    # Developer smoke of the N > 1 code path on a box with fewer GPUs than ranks: PG_BENCH_SHARE_GPU=1 puts
    # every rank on cuda:0 and rendezvous over gloo (RCCL refuses two ranks on one device).  Never set by the
    # driver; the real path is one rank per GPU over RCCL.
    share_gpu = world > 1 and os.environ.get("PG_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        stream = torch.cuda.current_stream().cuda_stream

    shard = world > 1 and args.mode == "shard"
    ctx = pa.Context(local_rank, stream if shard else None)
    from pairec_amd.dist import shard_range, sharded_step, GpuShardEngine
    if shard:
        begin, end = shard_range(args.rows * world, world, rank)       # N x rows table, one range per rank
    else:
        begin, end = 0, args.rows                                      # full replica
    table = pa.Table(ctx, end - begin, args.dim, row_offset=begin)
    table.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    prec = pa.PREC_BF16 if args.prec == "bf16" else pa.PREC_F32
    model = pa.RankModel(ctx, pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    expr = pa.Expr(RANK_EXPR)

    total_steps = args.warmup + args.steps
    # replica mode: every rank serves different users
    qs = [make_queries(o, s * (1 if shard else world) + (0 if shard else rank), R, args.dim) for s in range(total_steps)]

    if not shard:
        pipe = Pipeline1(pa, ctx, table, model, expr, R, K)
        d_qs = [ctx.to_device(q) for q in qs]

        def run(s):
            pipe.step(d_qs[s])

        def sync():
            ctx.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
    else:
        eng = GpuShardEngine(torch, ctx, table, model, expr, K, R)
        dev = torch.device("cuda", local_rank)
        t_qs = [torch.from_numpy(q).to(dev) for q in qs]

        def run(s):
            sharded_step(eng, dist, torch, t_qs[s], R, K)

        def sync():
            dist.barrier()
            torch.cuda.synchronize()

    # measured streaming-read ceiling of this GPU's HBM (plain read-only kernel over the same table)
    measured_gbs = table.hbm_read_probe(3)
    for s in range(args.warmup):
        run(s)
    sync()
    scan_ms, scan_launches, stage = [], 0, {"recall": [], "rank": []}
    import gc
    gc.collect()
    gc.disable()                                      # no collector pauses inside the timed region
    t0 = time.perf_counter()
    step_wall = []
    for s in range(args.warmup, total_steps):
        ts = time.perf_counter()
        run(s)
        ms, nbytes = ctx.last_scan_kernel()          # HIP events around the scan launches (this step)
        scan_ms.append(ms)
        step_wall.append((time.perf_counter() - ts) * 1e3)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("PG_BENCH_STEPTIMES"):          # developer aid: per-step wall and scan times
        print("step wall ms:", " ".join("%.2f" % x for x in step_wall), "| scan ms:", " ".join("%.2f" % x for x in scan_ms),
              "| rescans", ctx.stats().recall_rescans, file=sys.stderr)
    st = ctx.stats()
    if world > 1:
        dev = torch.device("cpu") if share_gpu else torch.device("cuda", local_rank)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = R * K * args.steps / elapsed * (1 if shard else world)
    # Algorithmic bytes of one table pass: the screen streams the table's shadow — int8 at dim 128 (rows x dim
    # x 1), bf16 at dim 64 — and the fp32 rows (rows x dim x 4, SURVEY.md 8d's figure for a scan
    # of the table itself) are only gathered for the ~1e-4 fraction of rows that reach the exact re-scoring.
    # Tables the screen cannot serve (dim > 128) are scanned in fp32.
    elem_bytes = table.screen_info()[0]
    screened = elem_bytes != 0
    shard_bytes = (end - begin) * args.dim * (elem_bytes if screened else 4)
    fp32_bytes = (end - begin) * args.dim * 4
    scan_avg_ms = float(np.mean(scan_ms))
    achieved = shard_bytes / (scan_avg_ms * 1e-3) / 1e9
    out = {
        "metric": "ranked items/sec, 5k-cand DNN rank (recall top-5000 of 100M x 128 -> DNN3 -> fuse -> sort)",
        "value": value, "unit": "ranked items/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
        "config": {"workload": "configs[2]: recall 5k of %dx%d fp32 table in HBM -> 3-layer DNN rank "
                               "(256-512-256-1, %s MFMA) -> RankScore fusion (fp64) -> ItemRankScore sort"
                               % (args.rows, args.dim, args.prec),
                   "requests_per_step": R, "candidates_per_request": K, "table_rows": args.rows,
                   "dim": args.dim,
                   "parallelism": ("table row-range shards x%d (%d rows total), all_gather top-K merge + all_reduce scores"
                                   % (world, args.rows * world)) if shard else
                                  ("request-parallel x%d, table replicated per GPU, no data-path collective" % world
                                   if world > 1 else "1 GPU")},
        "roofline": {"bound": "hbm", "kernel": scan_kernel_name(R, args.dim, elem_bytes), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(R),
                     "measured_peak": measured_gbs, "frac_of_measured": achieved / measured_gbs,
                     "bytes_per_pass": shard_bytes, "ms_per_pass": scan_avg_ms,
                     "mfma_flops_per_pass": 2.0 * (end - begin) * args.dim * R if screened else None,
                     "mfma_frac": (2.0 * (end - begin) * args.dim * R / (scan_avg_ms * 1e-3) / 1e12 /
                                   (MFMA_I8_PEAK_TOPS if elem_bytes == 1 else MFMA_BF16_PEAK_TFLOPS))
                     if screened else None,
                     "shadow_elem_bytes": elem_bytes,
                     "fp32_table_bytes": fp32_bytes,
                     "fp32_table_equivalent_gbs": fp32_bytes / (scan_avg_ms * 1e-3) / 1e9,
                     "note": "algorithmic bytes = shard rows x dim x shadow_elem_bytes per table pass: the pass streams the "
                             "int8 (dim 128) / bf16 (dim 64) shadow of the fp32 table, an exact integer / rigorous bound, with "
                             "exact fp32 re-scoring of the ~1e-4 of rows that pass it; one pass serves %d requests; "
                             "duration = sum of the pass's scan-stage launches (exact seed of the pilot sample, screened sample launch, "
                             "screened full pass, exact re-scoring), HIP events on the launch stream" % R},
        "stages_ms": {"recall_device_ms": st.last_recall_ms, "rank_device_ms": st.last_rank_ms},
        "rank_roofline": {"bound": "mfma", "kernel": "pg::dnn3_ws_kernel",
                          "achieved": (R * K / (world if shard else 1)) * FLOPS_PER_ITEM / max(st.last_rank_ms, 1e-9) / 1e9,
                          "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": (R * K / (world if shard else 1)) * FLOPS_PER_ITEM / max(st.last_rank_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS}
        if args.prec == "bf16" and st.last_rank_ms > 0 else None,
    }

    if world == 1 and args.latency_reqs > 0:  # noqa: E129
        # p50 single-request latency (R=1), same pipeline, inputs resident
        lat = []
        lq = [ctx.to_device(make_queries(o, 7000 + i, 1, args.dim)) for i in range(min(args.latency_reqs, 1000))]
        for i in range(args.latency_reqs):
            dq = lq[i % len(lq)]                      # distinct users
            ctx.synchronize()
            t1 = time.perf_counter()
            pipe.step(dq, R=1)
            ctx.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = lat[len(lat) // 10:]
        out["p99_request_latency_ms"] = float(np.percentile(lat, 99))
        out["p50_request_latency_ms"] = float(np.median(lat))

    if rank == 0:
        out["device"] = device_info()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(o, args, R, K)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
