#!/usr/bin/env python3
"""bench.py — ranked items/sec of pairec's recall → rank → sort hot path on MI355X.

Workload (BASELINE.json configs[1]+[2], the configuration the metric is quoted on): a batch of R=256 requests;
each request = exact inner-product recall of the top 5000 of a 100M x 128 fp32 item table resident in HBM →
3-layer DNN rank (256→512→256→1, bf16 MFMA) of those 5000 candidates → RankScore fusion in fp64 → ItemRankScore
(descending) sort.  A "step" is one such batch; value = ranked items / s (R*5000 per step), inputs resident in HBM
when the timed region starts.  Steps are issued through pg_recommend_dnn3_begin / pg_recommend_end, two batches
deep and alternating between two library contexts (two HIP streams with their own scratch), so the latency-bound head
and tail of one batch run under the other's scan / rank kernels (each batch is verified when it is ended).

  python bench.py --gpus N --steps K --warmup W
N > 1, one rank per GPU over RCCL: under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started plainly
(`python bench.py --gpus N`), it starts its N ranks itself through torch.distributed.run before anything touches the GPU, and
exits non-zero when the box has fewer than N GPUs (PG_BENCH_SHARE_GPU=1: developer mode, all ranks on cuda:0 over gloo with
host-staged exchanges; the line then says n_gpus 1, ranks N).  Two one-process-per-GPU modes:
  --mode replica (default): the 51.2 GB table fits one GPU, so it is replicated and the *requests*
      are sharded — every rank runs its own batches, no data-path collective, weak scaling.
  --mode shard: cfg-5 style — every rank holds --rows rows of one N x --rows table (contiguous row
      ranges), all_gather of the per-shard top-K lists + all_reduce of the score slab per step, DPP on the merged
      top-500 (pairec_amd/dist.py); value counts each request once.
and two ONE-process modes over N devices, no torch — what a cgo host calls (csrc/group.hip):
  --mode group: cfg-5 in one process: pg_group_* over N devices (peer stores + HIP events instead of collectives), two steps
      in flight, DPPSort on the merged top-500, host buffers in and out.
  --mode router: cfg-2/3 replicas in one process: one coalescer per device behind pg_router_*, 768 x N host threads with one
      request outstanding each; the timed region is exactly steps x 256 x N requests.

At N = 1 the same JSON line also carries (each a bounded, separately timed leg after the headline region):
  "concurrent_callers"  the same table / model / k served through the request coalescer to --callers host threads
                        that each issue ONE request at a time (how pairec calls its plug-ins), items/s + p50/p99
  "gaussian_table"      the headline measurement repeated on N(0,1) rows (the int8 screen's less favourable case)
  "other_configs"       cfg 1 (1M x 64, top-200, ascending ItemScore sort), cfg 4 (FM + two-tower rank, 1M-row field
                        tables) with their own rooflines, and one shard of cfg 5 (125M rows, recall + rank + DPP)
  "cpu_baseline"        the oracle (C port of the reference-shaped CPU path) on a bounded sample, all host cores

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# The oracle (the checker and the cpu_baseline leg) is an OpenMP library: its worker threads spin for a while after every
# parallel region, on the cores the HIP runtime's own threads want — single-request latencies measured right after the oracle
# made the queries picked up 10-100 ms outliers from that.  Passive waiting keeps the checker out of the measurement.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 (v_mfma_i32_32x32x32_i8 issues at the bf16 rate with twice the k)
FLOPS_PER_ITEM = 524800        # SURVEY.md §8(d) cfg 3: 2*(256*512+512*256+256)
FM2T_BYTES_PER_ITEM = 544      # SURVEY.md §8(d) cfg 4: 8 ids x 4 B + 8 rows x 64 B
FM2T_FLOPS_PER_ITEM = 99456
RANK_EXPR = "${gpu_dnn}*(1+${current_score})^0.1"      # RankConf.RankScore: model score x recall score
PROFILE_JSON = os.path.join(ROOT, "profiles", "r3_scan_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--calibrate", type=int, default=10,
                    help="untimed batches served before the warm-up steps, like the shadow build: the table's threshold "
                         "model (DESIGN.md 4.1, plan 0) starts predicting once it has observed 1024 verified queries of one K")
    ap.add_argument("--rows", type=int, default=None,
                    help="table rows PER GPU: default 100M (replica / router: the whole table on every GPU), 125M for shard / "
                         "group (configs[4]: 1 B rows over 8 GPUs)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=256, help="requests per step (<= 256 = one table pass)")
    ap.add_argument("--prec", choices=["bf16", "f32", "bf16x3"], default="bf16x3",
                    help="rank model precision of the headline: bf16x3 (default) = split bf16 — configs[2]'s bf16 MFMA with hi + lo "
                         "operands, the one matrix-pipe mode whose scores are within north_star's 1e-5 of the fp32 path; bf16 = plain "
                         "bf16 operands (faster, 4e-5 from fp32); f32 = fp32 MFMA")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the requests-per-pass sweep (batch_sweep)")
    ap.add_argument("--no-clustered", action="store_true", help="skip the clustered-table leg (clustered_table)")
    ap.add_argument("--mode", choices=["replica", "shard", "group", "router"], default="replica")
    ap.add_argument("--table-dist", choices=["uniform", "gaussian"], default="uniform",
                    help="headline table: SURVEY.md 8d's normalised uniform rows, or N(0,1) rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline measurement only (profiling runs)")
    ap.add_argument("--no-power", action="store_true", help="skip the power / clock readings (rocm-smi sampled beside three short loops)")
    ap.add_argument("--no-preflight", action="store_true",
                    help="N > 1: skip the parity preflight of the sharded step on the run's own wire (it runs before anything is timed)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs behind roofline.traffic")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the PG_PREC_F32 headline and the bf16-vs-f32 figures")
    ap.add_argument("--no-rank-shapes", action="store_true", help="skip the rank-stage-alone leg (DNN3 per hidden shape)")
    ap.add_argument("--contexts", type=int, default=2,
                    help="library contexts (= HIP streams with their own scratch) the batches alternate between")
    ap.add_argument("--callers", type=int, default=768, help="host threads of the concurrent-callers leg (0 = skip)")
    ap.add_argument("--callers-seconds", type=float, default=4.0)
    ap.add_argument("--page", type=int, default=100, help="entries each concurrent caller asks for (ctx.Size)")
    ap.add_argument("--latency-reqs", type=int, default=500,
                    help="single-request latency samples at N=1 (the first 10 %% are discarded as warm-up, SURVEY.md 8d)")
    args = ap.parse_args()
    if args.rows is None:
        args.rows = 125_000_000 if args.mode in ("shard", "group") else 100_000_000
    return args


def free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    p_ = so.getsockname()[1]
    so.close()
    return p_


def count_gpus():
    """GPUs of this box WITHOUT initialising the HIP runtime in this process: the KFD topology in sysfs (a node with SIMDs is
    a GPU); where that is unreadable, a short-lived child asks torch.  (torch.cuda.device_count() may fall back to
    hipGetDeviceCount, which does initialise the runtime — ADVICE r4.)"""
    import glob
    import subprocess
    n, seen = 0, False
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(prop):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count":
                    seen = True
                    n += int(f[1]) > 0
        except OSError:
            pass
    if seen:
        # (the topology ignores HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, which the ranks honour: cap by
        # the shortest of the lists that are set — ADVICE r5)
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v is not None:
                n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
        return n
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, the driver's own
    command line) and exit with their code.  This process never initialises the GPU (count_gpus reads sysfs or asks a
    child); the ranks are always fresh children, never an exec of this process.  Fewer GPUs than ranks is an error, not a
    quiet one-rank run."""
    import subprocess
    have = count_gpus()
    share = os.environ.get("PG_BENCH_SHARE_GPU") == "1"
    if have < args.gpus and not share:
        sys.stderr.write("bench.py: --gpus %d but this box has %d GPU(s); refusing to run fewer ranks than asked for "
                         "(PG_BENCH_SHARE_GPU=1 runs the %d ranks on cuda:0 as a developer check, never a measurement)\n"
                         % (args.gpus, have, args.gpus))
        raise SystemExit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def profile_numbers(R):
    """What the committed rocprofv3 PMC passes of this same command measured for the dominant kernel at this batch
    size (profiles/r3_scan_traffic.json, written by scripts/make_traffic_json_r3.py): HBM bytes per table pass
    (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md §HBM, + WRITE_SIZE) and the MFMA pipe's busy
    fraction.  NOT measured in this run — hence the field name traffic_from_profile."""
    try:
        # round 6: the 1 … 64-request passes (4-bit shadow through the matrix pipe) have summaries of their own (scripts/profile_r6.sh)
        p6 = os.path.join(ROOT, "profiles", "r6_sweep_%d_summary.json" % R)
        if os.path.exists(p6):
            with open(p6) as f:
                pm = json.load(f).get("pmc", {})
            tot = sum(v.get("fetch_bytes", 0) + v.get("write_bytes", 0) for v in pm.values())
            busy = max([v.get("mfma_busy_frac", 0.0) for v in pm.values()] or [0.0])
            if tot > 0:
                return {"hbm_bytes_per_pass": int(tot), "mfma_busy_frac": round(busy, 4), "source": os.path.relpath(p6, ROOT)}
        with open(PROFILE_JSON) as f:
            d = json.load(f)
        e = d.get(str(R), {})
        return {"hbm_bytes_per_pass": e.get("hbm_bytes_per_pass"), "mfma_busy_frac": e.get("mfma_busy_frac"),
                "source": os.path.relpath(PROFILE_JSON, ROOT)} if e else None
    except Exception:
        return None


def measure_traffic_live(args, R):
    """HBM bytes per table pass of the scan stage, measured in THIS run: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE — one counter per run, no trace domains, the program directly behind `--`),
    steady state = the last dispatches of the run, FETCH_SIZE (KB) x 1024 x 2 per the gfx950 correction of
    MI355X_MICROARCH.md, WRITE_SIZE (KB) x 1024.  Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ):
        return None, "this run is itself being profiled: no nested rocprofv3"
    kernels = ("screen_kernel", "screen4_kernel", "screen4m_kernel", "rescreen8_kernel", "screen_decode_kernel", "rescore_kernel")
    steps = 6
    detail = {}
    total = 0.0
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        d = tempfile.mkdtemp(prefix="pg_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", PG_BENCH_CHILD="1")
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   "python3", os.path.abspath(__file__), "--steps", str(steps), "--warmup", "2", "--batch", str(R),
                   "--rows", str(args.rows), "--dim", str(args.dim), "--k", str(args.k), "--prec", args.prec,
                   "--table-dist", args.table_dist, "--calibrate", str(args.calibrate),
                   "--no-cpu-baseline", "--latency-reqs", "0", "--no-extras", "--no-rank-shapes", "--no-f32-leg", "--no-batch-sweep", "--no-clustered", "--contexts", "1", "--callers", "0"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (counter, r.returncode)
            by = {}
            for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(p)):
                    kn = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") != counter:
                        continue
                    for k_ in kernels:
                        if k_ in kn:
                            by.setdefault(k_, []).append((int(row.get("Dispatch_Id", 0)), float(row.get("Counter_Value", 0) or 0)))
                            break
            if not by:
                return None, "no %s rows for the scan-stage kernels" % counter
            for k_, v in by.items():
                v.sort()
                # the timed steps' passes are the last dispatches (one screened launch per pass in the steady state)
                tail = [x[1] for x in v[-steps:]]
                b = sum(tail) / len(tail) * scale
                detail["%s_%s_bytes" % (k_, counter.lower())] = int(b)
                total += b
        except Exception as e:                      # noqa: BLE001 — the measurement is optional
            return None, "%s: %s" % (type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(total), detail


def measure_kernel_traffic(script_args, kernel_substr, tail=8):
    """HBM bytes per launch of one kernel, measured now: two child runs of a small driver script under `rocprofv3 --pmc`
    (FETCH_SIZE, WRITE_SIZE — one counter per run, no trace domains, the program directly behind `--`); FETCH_SIZE (KB) x 1024
    x 2 per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE (KB) x 1024; mean over the kernel's last `tail`
    dispatches.  Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ):
        return None, "this run is itself being profiled: no nested rocprofv3"
    total, detail = 0.0, {}
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        d = tempfile.mkdtemp(prefix="pg_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", "python3"] + script_args
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (counter, r.returncode)
            v = []
            for p_ in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(p_)):
                    if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                        v.append((int(row.get("Dispatch_Id", 0)), float(row.get("Counter_Value", 0) or 0)))
            if not v:
                return None, "no %s rows for %s" % (counter, kernel_substr)
            v.sort()
            t_ = [x[1] for x in v[-tail:]]
            b = sum(t_) / len(t_) * scale
            detail["%s_bytes" % counter.lower()] = int(b)
            total += b
        except Exception as e:                      # noqa: BLE001 — the measurement is optional
            return None, "%s: %s" % (type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(total), detail


def scan_kernel_name(R, dim, elem_bytes):
    """The kernel recall.hip dispatches for the full-table pass at this batch size (dispatch_screen):
    screen_kernel<DIM, NQB, WAVES, SPLIT, VAR, I8, QH, L2> (L2 = 0: inner product)."""
    if elem_bytes == 0:
        return "pg::scan_kernel<%d,1>" % dim
    if elem_bytes == 1:
        if R > 128:
            return "pg::screen_kernel<128,4,8,1,0,true,2,0>"
        nqb = 4 if R > 64 else (2 if R > 32 else 1)
        return "pg::screen_kernel<128,%d,8,1,0,true,1,0>" % nqb
    nqb, waves = (8, 4) if R > 128 else ((4, 8) if R > 64 else ((2, 8) if R > 32 else (1, 8)))
    return "pg::screen_kernel<%d,%d,%d,1,0,false,1,0>" % (dim, nqb, waves)


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


def parse_smi(txt):
    """package power (W), shader clock (MHz) and the power cap (W) of GPU 0 out of `rocm-smi --showpower --showclocks --showmaxpower`."""
    import re
    pw = re.search(r"GPU\[0\].*?Current Socket Graphics Package Power \(W\): ([0-9.]+)", txt)
    ck = re.search(r"GPU\[0\].*?sclk clock level: \S+ \(([0-9.]+)Mhz\)", txt)
    cap = re.search(r"GPU\[0\].*?Max Graphics Package Power \(W\): ([0-9.]+)", txt)
    if not pw or not ck:
        return None
    return {"package_w": float(pw.group(1)), "sclk_mhz": float(ck.group(1)), "cap_w": float(cap.group(1)) if cap else None}


def smi_sample():
    """One reading of rocm-smi (a child process).  None when the tool is missing or prints something else."""
    import subprocess
    try:
        return parse_smi(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True,
                                        text=True, timeout=20).stdout)
    except Exception:
        return None


def power_leg(step_fn, drain_fn, seconds=3.0, sync_each=False):
    """What the package draws and clocks at while `step_fn` runs back to back for `seconds`: rocm-smi sampled from a thread (the
    tool is a child process per reading, ~0.3 s each; the first reading is dropped — the loop may still be ramping).  The two
    kernels that dominate the headline step run AT the package power limit with the shader clock pulled below its 2.4 GHz
    maximum — which is why their matrix-pipe busy fraction x clock, not the busy fraction alone, is what schedules can move."""
    import threading
    stop, readings = threading.Event(), []

    def sampler():
        while not stop.is_set():
            r_ = smi_sample()
            if r_ is None:
                return
            readings.append(r_)
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    n = 0
    th.start()
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            step_fn()
        n += 16
        if sync_each:                                  # (asynchronous calls: do not run ahead of the clock)
            drain_fn()
    drain_fn()
    el = time.perf_counter() - t0
    stop.set()
    th.join(timeout=30)
    rd = readings[1:-1] if len(readings) > 3 else readings
    if not rd:
        return None
    return {"seconds": el, "steps": n, "ms_per_step": el / n * 1e3, "readings": len(rd),
            "package_w": float(np.median([r_["package_w"] for r_ in rd])), "sclk_mhz": float(np.median([r_["sclk_mhz"] for r_ in rd])),
            "cap_w": rd[0]["cap_w"]}


def make_queries(o, step, R, dim):
    # 1000 distinct users cycle through the run
    return o.synth_rows(o.SEED_QUERY, (step * R) % 1000, R, dim)


# ------------------------------------------------------------------------------------------------
# single-GPU pipeline on raw device pointers (no torch on this path)
# ------------------------------------------------------------------------------------------------
class Pipeline1:
    """pg_recommend_dnn3_begin / _end with `depth` output-buffer sets: step s + 1 is enqueued before step s is
    ended, so the device always has the next batch queued behind the running one."""

    def __init__(self, pa, ctx, table, model, expr, R, K, depth=2, extra_ctxs=()):
        self.pa, self.ctx, self.table, self.model, self.expr, self.R, self.K = pa, ctx, table, model, expr, R, K
        n = R * K
        m = ctx.malloc
        self.ctxs = [ctx] + list(extra_ctxs)     # batches alternate between the contexts (own stream + scratch each)
        depth = max(depth, len(self.ctxs))
        self.bufs = [(m(n * 8), m(n * 4), m(n * 4), m(n * 8), m(n * 4)) for _ in range(depth)]
        self.depth = depth
        self.inflight = []          # (context, ticket), oldest first
        self.issued = 0
        self.scan_ms = []

    def begin(self, d_q, R=None):
        from pairec_amd import _lib
        ctx = self.ctxs[self.issued % len(self.ctxs)]
        b = self.bufs[self.issued % self.depth]
        tk = C.c_void_p()
        _lib.check(ctx.L.pg_recommend_dnn3_begin(ctx.h, self.table.h, self.model.h, self.expr.h, b"gpu_dnn", d_q,
                                                 R or self.R, self.K, b[0], b[1], b[2], b[3], b[4], None, C.byref(tk)))
        self.inflight.append((ctx, tk))
        self.issued += 1

    def end_oldest(self):
        from pairec_amd import _lib
        ms = C.c_double()
        ctx, tk = self.inflight.pop(0)
        _lib.check(ctx.L.pg_recommend_end(ctx.h, tk, C.byref(ms)))
        self.scan_ms.append(ms.value)

    def step(self, d_q, R=None):
        """Keep `depth` batches in flight: enqueue this step, end the one issued depth - 1 steps ago."""
        self.begin(d_q, R)
        while len(self.inflight) >= self.depth:
            self.end_oldest()

    def drain(self):
        while self.inflight:
            self.end_oldest()


def cpu_baseline(o, args, R, K):
    """The oracle (a C port of the reference-shaped CPU path: scan + per-worker heap top-K, DNN forward in batches,
    sort) on a bounded sample of the same workload.  kind = "port": the reference is Go with its arithmetic in remote
    services; nothing of it can run here.

    Recall sample: in a full run every one of the host's C cores scans rows/C rows against all R queries and keeps its
    own K-heaps.  The sample runs min(C, 16) threads on exactly that share each (a slice of min(C, 16) x rows/C rows), so
    its wall time IS the estimate of the full pass with all cores busy (the final per-query merge of the workers' heaps,
    ~0.1 s, is not included).  Scaling a small slice linearly by rows — round 1 — overstated the heap work 5x: a worker
    that sees 31 K rows puts 16 % of them through its heaps, one that sees 390 K rows 1 %."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = max(args.rows // cores, K)
    ts = min(cores, 16)
    slice_rows = min(share * ts, 8_000_000)
    tab = o.synth_rows(o.SEED_TABLE, 0, slice_rows, args.dim)
    q = make_queries(o, 0, R, args.dim)
    t0 = time.time()
    rows, scores = o.recall_topk(tab, q, K, threads=ts)
    t_recall_slice = time.time() - t0
    t_recall = t_recall_slice * (share * ts / slice_rows)
    w = o.Dnn3Weights()
    n_sample = max(20000, min(R * K, cores * 4000))        # enough items per worker that start-up does not dominate
    cand = tab[rows[0][:K].astype(np.int64) % slice_rows]
    items = np.tile(cand, (n_sample // K + 1, 1))[:n_sample]
    # the blocked fp32 MLP (4 items per weight pass, AVX2 fma) on all cores; the thread pool is warmed by a first call, and
    # the better of {all cores, half of them} is kept (SMT siblings share the FMA pipes)
    prec_i = 1 if args.prec == "bf16" else 0
    o.dnn3_forward(w, prec_i, q[0], items[:cores * 8], threads=cores)
    t_rank, rank_threads = None, cores
    for th in sorted({cores, max(1, cores // 2)}, reverse=True):
        t0 = time.time()
        sc = o.dnn3_forward(w, prec_i, q[0], items, threads=th)
        dt = time.time() - t0
        if t_rank is None or dt < t_rank:
            t_rank, rank_threads = dt, th
    rank_gflops = n_sample * FLOPS_PER_ITEM / t_rank / 1e9
    t_rank *= R * K / n_sample
    t0 = time.time()
    for r in range(R):
        o.sort_scores(sc[:K].astype(np.float64), True)
    t_sort = time.time() - t0
    total = t_recall + t_rank + t_sort
    return {
        "value": R * K / total, "unit": "ranked items/s", "cores": cores, "kind": "port",
        "sample": "recall: %d of the %d cores' shares of the table (%d rows each = a %d-row slice) x %d queries: %.2f s = "
                  "the full pass with every core on its share; rank: %d items on all cores (scaled to %d); sort: %d x %d"
                  % (ts, cores, share, slice_rows, R, t_recall_slice, n_sample, R * K, R, K),
        "stage_seconds_per_step": {"recall": t_recall, "rank": t_rank, "sort": t_sort},
        "rank_leg": {"threads": rank_threads, "gflops": rank_gflops, "items_per_weight_pass": 4},
    }


def roofline_block(table, R, args, rows_local, scan_avg_ms, measured_gbs, scan_bytes=None):
    """Both denominators, side by side: `frac` prices the bytes the pass actually streams (the int8 / bf16 shadow),
    `frac_survey_8d` prices SURVEY.md 8(d)'s algorithmic figure — the fp32 table, rows x dim x 4 — and exceeds 1
    whenever the screen is on, because the fp32 rows are never streamed (only the ~1e-4 of rows that pass the exact
    integer / rigorous bound are gathered for re-scoring; the answers are bit-identical).  `bound` names the
    resource with the larger utilisation (HBM fraction of the streamed bytes vs the matrix pipe)."""
    elem_bytes = table.screen_info()[0]
    screened = elem_bytes != 0
    shard_bytes = rows_local * args.dim * (elem_bytes if screened else 4)
    # batches of <= 64 queries: the full pass streams the 4-bit shadow (68 B per row, csrc/recall_i4m.hip) — the engine's own
    # byte count of the last recall's scan launches (pilot sample on the main shadow + that pass) says whether it did
    four_bit = bool(screened and R <= 64 and args.dim == 128 and scan_bytes and scan_bytes < shard_bytes)
    if four_bit:
        shard_bytes = int(scan_bytes)
    fp32_bytes = rows_local * args.dim * 4
    t = scan_avg_ms * 1e-3
    achieved = shard_bytes / t / 1e9
    mfma_ops = 2.0 * rows_local * args.dim * R
    mfma_peak = MFMA_I8_PEAK_TOPS if elem_bytes == 1 else (MFMA_BF16_PEAK_TFLOPS if elem_bytes == 2 else 157.3)
    mfma_frac = mfma_ops / t / 1e12 / mfma_peak
    prof = profile_numbers(R)
    busy = (prof or {}).get("mfma_busy_frac") or 0.0
    hbm_frac = achieved / HBM_PEAK_GBS
    return {
        "bound": "mfma" if (max(mfma_frac, busy) > hbm_frac and not four_bit) else "hbm",
        "kernel": ("pg::screen4m_kernel<%d> (+ rescreen8_kernel, rescore_kernel)" % (1 if R <= 32 else 2))
                  if four_bit else scan_kernel_name(R, args.dim, elem_bytes),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
        "frac_basis": "achieved / peak / unit / frac price the bytes the pass streams against HBM whatever `bound` says (the "
                      "series is comparable across rounds); the matrix-pipe side of the same pass is mfma_achieved / mfma_peak "
                      "/ mfma_frac (algorithmic int8 ops on the nominal dense peak) and traffic_from_profile.mfma_busy_frac",
        "frac_survey_8d": fp32_bytes / t / 1e9 / HBM_PEAK_GBS,
        "traffic": None, "traffic_from_profile": prof,
        "measured_peak": measured_gbs, "frac_of_measured": achieved / measured_gbs if measured_gbs else None,
        "bytes_per_pass": shard_bytes, "fp32_table_bytes": fp32_bytes, "ms_per_pass": scan_avg_ms,
        "shadow_elem_bytes": elem_bytes,
        "mfma_ops_per_pass": mfma_ops, "mfma_achieved": mfma_ops / t / 1e12, "mfma_peak": mfma_peak,
        "mfma_unit": "Top/s" if elem_bytes == 1 else "TFLOP/s", "mfma_frac": mfma_frac,
        "note": "bytes_per_pass = rows x dim x shadow_elem_bytes: the pass streams the int8 (dim 128) / bf16 shadow of the "
                "fp32 table — an exact integer / rigorous bound of every score — and re-scores the survivors exactly in "
                "fp32; frac_survey_8d > 1 says exactly that (the 51.2 GB fp32 table is not streamed).  One pass serves "
                "%d requests; ms_per_pass = sum of the pass's scan-stage launches (pilot seed, pilot sample launch, "
                "full pass, exact re-scoring), HIP events on the launch stream, measured with ONE batch in flight (the "
                "headline region overlaps two batches on two streams, which inflates per-kernel event times).  traffic = HBM bytes per pass of "
                "these launches from two child runs of this script under rocprofv3 --pmc (FETCH_SIZE x 2 per the gfx950 correction, + "
                "WRITE_SIZE; null when rocprofv3 is unavailable); traffic_from_profile quotes the committed PMC passes of the same command." % R,
    }


def run_headline(pa, ctx, table, model, expr, qs_dev, args, R, K, sync, extra_ctxs=()):
    """warmup + timed steps; returns (elapsed seconds, per-step scan ms)."""
    pipe = Pipeline1(pa, ctx, table, model, expr, R, K, extra_ctxs=extra_ctxs)
    for s in range(args.warmup):
        pipe.step(qs_dev[s % len(qs_dev)])
    pipe.drain()
    sync()
    pipe.scan_ms = []
    import gc
    gc.collect()
    gc.disable()                                      # no collector pauses inside the timed region
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        pipe.step(qs_dev[s % len(qs_dev)])
    pipe.drain()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    return pipe, elapsed, pipe.scan_ms


def headline_spot_check(o, pipe, table, w, prec, queries_of_last_step, K):
    """One request of the LAST timed batch against the oracle (the checker, outside the timed region): the recall scores of
    the returned rows bit for bit (the oracle's dot products of the rows gathered from the table), the model scores within
    the precision mode's tolerance, the fused scores to 1e-12 relative and the ItemRankScore order = the oracle's sort of
    the device's own fused scores."""
    b = pipe.bufs[(pipe.issued - 1) % pipe.depth]
    ctx = pipe.ctx
    rows, rec, rnk = np.zeros(K, np.uint64), np.zeros(K, np.float32), np.zeros(K, np.float32)
    fus, order = np.zeros(K, np.float64), np.zeros(K, np.uint32)
    for a_, p_ in ((rows, b[0]), (rec, b[1]), (rnk, b[2]), (fus, b[3]), (order, b[4])):
        ctx.d2h(a_, p_)
    q0 = queries_of_last_step[0]
    emb = table.gather(rows.astype(np.uint32))
    want_rec = o.dot_scores(emb, q0[None])[0]
    want_rnk = o.dnn3_forward(w, 1 if prec == "bf16" else 0, q0, emb)    # bf16x3 is checked against the fp32 specification
    want_fus = o.widen_f32(rnk) * (1 + o.widen_f32(rec)) ** 0.1
    tol = {"bf16": 1e-5, "bf16x3": 1e-6}.get(prec, 2e-7)
    err_fp32 = float(np.max(np.abs(rnk.astype(np.float64) - (want_rnk if prec != "bf16" else o.dnn3_forward(w, 0, q0, emb)))))
    res = {"request": "request 0 of the last timed batch, %d candidates" % K,
           "rank_oracle": "fp32 specification (no rounding point mirrored)" if prec != "bf16" else "oracle that rounds operands to bf16 where the kernel does",
           "rank_max_abs_err_vs_fp32_oracle": err_fp32, "north_star_tolerance": 1e-5,
           "within_north_star_tolerance_of_fp32_oracle": bool(err_fp32 <= 1e-5),
           "recall_scores_bit_exact": bool(np.array_equal(rec.view(np.uint32), want_rec.view(np.uint32))),
           "recall_sorted": bool(np.all(np.diff(rec.astype(np.float64)) <= 0)),
           "rank_max_abs_err": float(np.max(np.abs(rnk.astype(np.float64) - want_rnk))), "rank_tolerance": tol,
           "fused_max_rel_err": float(np.max(np.abs(fus - want_fus) / np.maximum(np.abs(want_fus), 1e-300))),
           "order_equals_oracle_sort": bool(np.array_equal(order, o.sort_scores(fus, True)))}
    res["ok"] = bool(res["recall_scores_bit_exact"] and res["recall_sorted"] and res["rank_max_abs_err"] <= tol and
                     res["fused_max_rel_err"] <= 1e-12 and res["order_equals_oracle_sort"])
    return res


class LoadgenResult(C.Structure):
    _fields_ = [("requests", C.c_uint64), ("errors", C.c_uint64), ("seconds", C.c_double), ("p50_ms", C.c_double),
                ("p90_ms", C.c_double), ("p99_ms", C.c_double), ("max_ms", C.c_double), ("mean_ms", C.c_double),
                ("checksum", C.c_uint64)]


class LoadgenSpec(C.Structure):
    _fields_ = [("mode", C.c_int), ("user_vecs", C.c_void_p), ("n_users", C.c_uint32), ("dim", C.c_uint32),
                ("k", C.c_uint32), ("top_n", C.c_uint32), ("user_field_ids", C.c_void_p), ("n_user_fields", C.c_uint32),
                ("cand_pool", C.c_void_p), ("pool_size", C.c_uint32), ("rank_items", C.c_uint32),
                ("rel_pool", C.c_void_p), ("dpp", C.c_void_p)]


_host_lib = None


def host_lib():
    global _host_lib
    if _host_lib is None:
        _host_lib = C.CDLL(os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
        _host_lib.ph_loadgen_run_ex.argtypes = [C.c_void_p, C.POINTER(LoadgenSpec), C.c_uint32, C.c_uint32, C.c_double,
                                                C.POINTER(LoadgenResult)]
    return _host_lib


def loadgen(co, spec, callers, seconds, warm, flavour, items_per_request):
    """`callers` host threads in closed loops of single-request calls (pairec_amd/host/loadgen.cpp) → a result row."""
    res = LoadgenResult()
    s0 = co.stats()
    b0, r0 = s0.batches[flavour], s0.requests[flavour]
    rc = host_lib().ph_loadgen_run_ex(co.h, C.byref(spec), callers, warm, seconds, C.byref(res))
    s1 = co.stats()
    if rc or res.errors:
        raise RuntimeError("load generator failed (rc %d, %d errors)" % (rc, res.errors))
    return {"callers": callers, "value": res.requests * items_per_request / res.seconds,
            "requests_per_s": res.requests / res.seconds, "p50_ms": res.p50_ms, "p90_ms": res.p90_ms, "p99_ms": res.p99_ms,
            "mean_ms": res.mean_ms, "avg_batch": (s1.requests[flavour] - r0) / max(s1.batches[flavour] - b0, 1),
            "replans": s1.replans}


DPP_CANDIDATES, DPP_WINDOW, DPP_ALPHA = 500, 10, 1.0       # cfg 5's DPPSort: top 500 by score, alpha 1, window 10


def dpp_batch_rate(pa, ctx, table, model, expr, users, R, K, page, steps=6):
    """The caller-made batch the coalesced recommend + DPP leg is compared with: R requests per call through the
    device-level ABI (pg_recommend_dnn3_dev → pg_dpp_candidates_dev → pg_gather_owned_rows_dev → pg_dpp_batch_dev),
    pages copied to the host, one batch at a time on one context."""
    from pairec_amd import _lib
    L, h = ctx.L, ctx.h
    n, Cn = R * K, DPP_CANDIDATES
    bufs = [ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(R * 4)]
    c_rows, c_rel, c_emb = ctx.malloc(R * Cn * 8), ctx.malloc(R * Cn * 8), ctx.malloc(R * Cn * table.dim * 4)
    picks, pcnt = ctx.malloc(R * page * 4), ctx.malloc(max(R, 256) * 4)
    h_picks = np.zeros((R, page), np.uint32)
    h_rows = np.zeros((R, K), np.uint64)

    def step(i):
        d_q = ctx.to_device(users[(i * R) % 512:(i * R) % 512 + R])
        _lib.check(L.pg_recommend_dnn3_dev(h, table.h, model.h, expr.h, b"gpu_dnn", d_q, R, K, *bufs))
        _lib.check(L.pg_dpp_candidates_dev(h, bufs[4], bufs[0], bufs[3], R, K, Cn, c_rows, c_rel))
        _lib.check(L.pg_gather_owned_rows_dev(h, table.h, c_rows, R * Cn, c_emb))
        _lib.check(L.pg_dpp_batch_dev(h, c_emb, c_rel, R, Cn, table.dim, DPP_ALPHA, page, DPP_WINDOW, 1, picks, pcnt))
        ctx.d2h(h_picks, picks)
        ctx.d2h(h_rows[:, :Cn].copy(), c_rows)
        ctx.free(d_q)
    step(0)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 1)
    dt = (time.perf_counter() - t0) / steps
    for p in bufs + [c_rows, c_rel, c_emb, picks, pcnt]:
        ctx.free(p)
    return {"ms_per_batch": dt * 1e3, "value": R * K / dt, "requests_per_s": R / dt}


def concurrent_callers_leg(pa, o, ctx, table, model, expr, args, K):
    """--callers host threads, each blocked in ONE pg_coalescer_recommend at a time (closed loop) — how pairec's
    goroutines call IAlgorithm.Run / Recall.GetCandidateItems.  The library forms the batches.  A second leg serves a
    [ItemRankScore, DPPSort] scene the same way: the DPP stage runs inside the coalesced call."""
    users = np.ascontiguousarray(o.synth_rows(o.SEED_QUERY, 0, 1000, args.dim))
    spec = LoadgenSpec(mode=0, user_vecs=users.ctypes.data, n_users=1000, dim=args.dim, k=K, top_n=args.page)
    out = {}
    co = pa.Coalescer(ctx, table, K, model, expr, "gpu_dnn", max_top_n=args.page, depth=3)
    try:
        main_leg = loadgen(co, spec, args.callers, args.callers_seconds, 2, 2, K)
        sweep = [loadgen(co, spec, c_, 1.5, 2, 2, K) for c_ in (8, 32, 128, 256, 512) if c_ != args.callers]
        solo = loadgen(co, spec, 1, 1.0, 3, 2, K)
        out = dict(main_leg)
        out.update({"other_caller_counts": sweep, "solo_caller": solo})
    except RuntimeError as e:
        out = {"error": str(e)}
    co.destroy()
    # the same scene with DPPSort behind the sort (cfg 5's stage: 500 candidates, page of ctx.Size, window 10)
    co = pa.Coalescer(ctx, table, K, expr=expr, algos=[("gpu_dnn", model)], max_top_n=args.page, depth=3,
                      dpp={"candidates": DPP_CANDIDATES, "alpha": DPP_ALPHA, "window": DPP_WINDOW})
    try:
        d_main = loadgen(co, spec, args.callers, max(2.0, args.callers_seconds / 2), 2, 2, K)
        d_solo = loadgen(co, spec, 1, 1.0, 3, 2, K)
        d_main.update({"solo_caller": d_solo, "vs_solo": d_main["value"] / d_solo["value"],
                       "caller_made_batch": dpp_batch_rate(pa, ctx, table, model, expr, users, 256, K, args.page)})
        d_main["vs_caller_made_batch"] = d_main["value"] / d_main["caller_made_batch"]["value"]
        d_main["note"] = ("recall -> DNN3 -> RankScore -> sort -> DPPSort(%d candidates, window %d) in ONE coalesced call per "
                          "request; the caller-made batch is the device-level ABI sequence for 256 requests, one batch at a time"
                          % (DPP_CANDIDATES, DPP_WINDOW))
        out["recommend_dpp"] = d_main
    except RuntimeError as e:
        out["recommend_dpp"] = {"error": str(e)}
    co.destroy()
    out.update({
        "mode": "concurrent_callers", "page": args.page, "k": K, "unit": "ranked items/s",
        "note": "closed loop: every caller has one request outstanding (a goroutine blocked in IAlgorithm.Run); results "
                "cross PCIe (the page: rows, three scores per entry); the coalescer batches up to 256 requests per table "
                "pass, 3 batches in flight",
    })
    return out



SWEEP_BATCHES = (1, 4, 8, 16, 32, 64, 128, 256)


def batch_sweep_leg(pa, o, ctx, table, model, expr, args, K, extra_ctxs, measured_gbs):
    """Requests per table pass, R in SWEEP_BATCHES (SURVEY.md 8(d)'s Qb in {1, 8, 32} / R in {1, 16, 64} are points of it): what a
    pass costs when per-request callers (service/recall/vector_recall.go:32-123, one call per request; rank_service.go:264-289)
    fill it only partly.  Per point: the scan stage's event-timed launches with ONE batch in flight (ms_per_pass: screen or scan
    kernels + the exact re-scoring), the bytes they streamed (the engine's own count: 68 B per row on the 4-bit shadow up to 64
    requests, 128 B on the int8 shadow beyond), frac = bytes / time / 8 TB/s, and the whole step (recall -> rank -> fusion ->
    sort) one batch at a time and two batches in flight as in the headline."""
    out = []
    for Rb in SWEEP_BATCHES:
        d = [ctx.to_device(make_queries(o, 3000 + 17 * i, Rb, args.dim)) for i in range(6)]
        one = Pipeline1(pa, ctx, table, model, expr, Rb, K, depth=1)
        for i in range(3):
            one.step(d[i])
        one.drain()
        ctx.synchronize()
        n = 12
        passes, nbytes = [], 0
        t0 = time.perf_counter()
        for i in range(n):
            one.begin(d[i % 6])
            one.drain()
            ms_, nbytes = ctx.last_scan_kernel()
            passes.append(ms_)
        ctx.synchronize()
        wall1 = (time.perf_counter() - t0) / n
        two = Pipeline1(pa, ctx, table, model, expr, Rb, K, extra_ctxs=extra_ctxs)
        for i in range(3):
            two.step(d[i])
        two.drain()
        for c_ in [ctx] + list(extra_ctxs):
            c_.synchronize()
        t0 = time.perf_counter()
        for i in range(2 * n):
            two.step(d[i % 6])
        two.drain()
        for c_ in [ctx] + list(extra_ctxs):
            c_.synchronize()
        wall2 = (time.perf_counter() - t0) / (2 * n)
        for pipe_ in (one, two):
            for b in pipe_.bufs:
                for p_ in b:
                    ctx.free(p_)
        for p_ in d:
            ctx.free(p_)
        ms_pass = float(np.mean(passes))
        out.append({"R": Rb, "ms_per_pass": ms_pass, "ms_per_pass_min": float(np.min(passes)), "bytes_streamed": int(nbytes),
                    "bytes_per_row": nbytes / table.rows,
                    "achieved_gbs": nbytes / ms_pass / 1e6, "frac": nbytes / ms_pass / 1e6 / HBM_PEAK_GBS,
                    "frac_of_measured": nbytes / ms_pass / 1e6 / measured_gbs if measured_gbs else None,
                    "ms_per_step_one_in_flight": wall1 * 1e3, "ms_per_step": wall2 * 1e3,
                    "items_per_s": Rb * K / wall2, "requests_per_s": Rb / wall2})
    return out


MIX_SEED, MIX_CENTRES = 0x5EED0007, 1000


def clustered_table_leg(pa, o, ctx, table, model, expr, args, R, K, sync, extra_ctxs, headline_ms):
    """The headline step on CLUSTERED rows (pg_table_fill_mixture: MIX_CENTRES centres on the unit sphere, within-cluster noise of
    norm sigma, normalised; queries are further points of the same mixture, so a query's top-K sits inside one cluster whose
    members score close to each other — the case the int8 bound's slack multiplies suspects in; uniform and i.i.d. Gaussian rows
    are its two best).  Per sigma: items/s, suspects per answer (what the int8 screen let through / K) and how many of them reached
    the exact fp32 re-scoring after the two-digit refinement stage such tables switch on (csrc/recall_r2.hip),
    scan-stage ms with one batch in flight, batches that fell back (screen overflow -> exact scan; failed plan -> re-run) and a
    slice of the device's rows regenerated by the oracle bit for bit."""
    import copy
    out = []
    for sigma in (0.3, 0.1, 0.03):
        table.fill_mixture(MIX_SEED, MIX_CENTRES, sigma)
        eb = table.screen_info()[0]
        sl = table.download(table.rows - 4096, 4096)
        slice_ok = bool(np.array_equal(sl.view(np.uint32), o.synth_mixture_rows(MIX_SEED, table.rows - 4096, 4096, args.dim, MIX_CENTRES, sigma).view(np.uint32)))
        qs = [ctx.to_device(o.synth_mixture_rows(MIX_SEED, 1000 * s_, R, args.dim, MIX_CENTRES, sigma, stream=1)) for s_ in range(args.calibrate + 6)]
        a0 = copy.copy(args)
        a0.warmup, a0.steps = args.calibrate, 0
        run_headline(pa, ctx, table, model, expr, qs[6:], a0, R, K, sync, extra_ctxs)
        ctxs = [ctx] + list(extra_ctxs)
        s0 = [c_.stats() for c_ in ctxs]
        a1 = copy.copy(args)
        a1.warmup, a1.steps = 2, 10
        pipe, el, _ = run_headline(pa, ctx, table, model, expr, qs[:6], a1, R, K, sync, extra_ctxs)
        s1 = [c_.stats() for c_ in ctxs]

        def delta(f):
            return sum(getattr(b_, f) - getattr(a_, f) for a_, b_ in zip(s0, s1))
        a2 = copy.copy(args)
        a2.warmup, a2.steps = 1, 6
        _, _, scan = run_headline(pa, ctx, table, model, expr, qs[:6], a2, R, K, sync)
        nq = max(delta("recall_suspect_queries"), 1)
        for b in pipe.bufs:
            for p_ in b:
                ctx.free(p_)
        for p_ in qs:
            ctx.free(p_)
        ms_step = el / a1.steps * 1e3
        out.append({"sigma": sigma, "centres": MIX_CENTRES, "shadow_elem_bytes": eb, "value": R * K * a1.steps / el, "unit": "ranked items/s",
                    "ms_per_step": ms_step, "vs_uniform_headline_ms": ms_step / headline_ms,
                    "suspects_per_answer": delta("recall_suspects") / nq / K,
                    "rescored_per_answer": delta("recall_rescored") / nq / K,
                    "scan_stage_ms_per_pass": float(np.mean(scan)),
                    "batches": a1.warmup + a1.steps, "batches_on_predicted_thresholds": int(delta("recall_predicted")),
                    "batches_re_run_after_a_failed_plan": int(delta("recall_rescans")),
                    "batches_that_fell_to_the_exact_scan": int(delta("recall_screen_overflows")),
                    "slice_matches_oracle": slice_ok})
    return out


RANK_SHAPES_JSON = os.path.join(ROOT, "profiles", "r3_rank_shapes_pmc.json")
RANK_SHAPES = ((128, 128), (256, 128), (256, 256), (512, 256), (1024, 512))


def rank_shapes_leg(pa, o, ctx, table, R, K):
    """The DNN3 rank stage ALONE (tile table + request partial + MLP kernel), bf16, per hidden shape: R requests x K
    random candidate rows of the resident table, nothing else on the device — the kernel's own duration, which the
    pipelined headline loop cannot show (there a rank kernel runs right behind a scan, at the clock the power cap leaves).  FLOP are
    SURVEY.md 8(d)'s 2*(256*h1 + h1*h2 + h2) per item; mfma_busy comes from the committed PMC pass when there is one."""
    n = table.rows
    rng = np.random.default_rng(5)
    nI = R * K
    cand = rng.integers(0, n, nI).astype(np.uint32)
    offs = (np.arange(R + 1) * K).astype(np.uint32)
    us = o.synth_rows(o.SEED_QUERY, 0, R, 128)
    d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
    d_out = ctx.malloc(nI * 4)
    pmc = {}
    try:
        with open(RANK_SHAPES_JSON) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        pass
    kernels = {(512, 256): "pg::dnn3_ws_kernel", (1024, 512): "pg::dnn3_ls_kernel"}
    out = []
    for h1, h2 in RANK_SHAPES:
        w = o.Dnn3Weights(d_user=128, d_item=128, h1=h1, h2=h2, seed=o.SEED_WEIGHTS ^ (h1 + h2))
        m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
        flop = 2 * (256 * h1 + h1 * h2 + h2)
        best = 1e9
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m.rank_dnn3_dev(table, d_u, d_c, d_o, R, nI, d_out)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        dev_ms = ctx.stats().last_rank_ms
        m.destroy()
        key = "%d-%d" % (h1, h2)
        out.append({"shape": "256-%d-%d-1" % (h1, h2), "kernel": kernels.get((h1, h2), "pg::dnn3_rs_kernel<%d, %d, ...>" % (h1, h2)),
                    "ms_per_%d_items" % nI: best * 1e3, "device_ms": dev_ms, "items_per_s": nI / best,
                    "achieved": nI * flop / best / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": nI * flop / best / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "gather_gbs": nI * 512 / best / 1e9,
                    "mfma_busy_from_profile": (pmc.get(key) or {}).get("mfma_busy")})
    # the benchmark's shape at PG_PREC_BF16X3 (dnn3_x3_kernel: three bf16 products per term, fp32 scores)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16X3, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    best = 1e9
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.rank_dnn3_dev(table, d_u, d_c, d_o, R, nI, d_out)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    dev_ms = ctx.stats().last_rank_ms
    m.destroy()
    out.append({"shape": "256-512-256-1", "prec": "bf16x3", "kernel": "pg::dnn3_x3_kernel<512, 256>",
                "ms_per_%d_items" % nI: best * 1e3, "device_ms": dev_ms, "items_per_s": nI / best,
                "achieved": nI * FLOPS_PER_ITEM / best / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": nI * FLOPS_PER_ITEM / best / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                "executed_flop_per_item": 3 * 393216, "executed_frac": nI * 3 * 393216 / best / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                "gather_gbs": nI * 512 / best / 1e9, "mfma_busy_from_profile": 0.61})
    for p_ in (d_u, d_c, d_o, d_out):
        ctx.free(p_)
    return out


def precision_figures(pa, ctx, table, expr, m_bf16, m_f32, q, K, page=100, tau_requests=32):
    """bf16 mode against PG_PREC_F32 on the SAME requests (same table, same recalled candidates): what the bf16 MFMA
    costs in score and in order.  north_star states 1e-5 on float scores against the reference's fp32 / fp64 CPU path
    (float32 model outputs widened, algorithm/eas/easyrec_response.go:479-483); PG_PREC_F32 meets it unconditionally
    (2e-7 against the oracle), the bf16 mode only against an oracle that rounds at the same points — this is the
    distance between the two modes themselves."""
    r16 = pa.recommend_dnn3(ctx, table, m_bf16, expr, "gpu_dnn", q, K)
    r32 = pa.recommend_dnn3(ctx, table, m_f32, expr, "gpu_dnn", q, K)
    assert np.array_equal(r16[0], r32[0])                      # the recall does not depend on the rank precision
    d = np.abs(r16[2].astype(np.float64) - r32[2].astype(np.float64)).reshape(-1)
    df = np.abs(r16[3] - r32[3]).reshape(-1)
    R = q.shape[0]
    same_page = same_set = same_ties = 0
    overlap = []
    for r in range(R):
        a, b = r16[0][r][r16[4][r][:page]], r32[0][r][r32[4][r][:page]]
        same_page += bool(np.array_equal(a, b))
        # the same order "up to ties": the first mode's page, read in the f32 mode's fused scores, never rises by more than
        # two ulps of a score (two fp32 evaluations that sum in different orders cannot agree more closely than that)
        fa = r32[3][r][r16[4][r][:page]]
        same_ties += bool(np.all(np.diff(fa) <= 2.4e-7 * np.abs(fa[:-1]))) and len(set(a.tolist()) & set(b.tolist())) == page
        inter = len(set(a.tolist()) & set(b.tolist()))
        same_set += inter == page
        overlap.append(inter / page)
    taus, taus_page = [], []
    try:
        from scipy.stats import kendalltau
        for r in range(0, R, max(1, R // tau_requests)):
            taus.append(float(kendalltau(r16[3][r], r32[3][r]).statistic))
            top = r32[4][r][:page]                             # the f32 page's items, as the bf16 mode orders them
            taus_page.append(float(kendalltau(r16[3][r][top], r32[3][r][top]).statistic))
    except Exception:                                          # noqa: BLE001 — scipy is optional here
        pass
    return {
        "items": int(d.size), "max_abs_dscore": float(d.max()), "p99_abs_dscore": float(np.percentile(d, 99)),
        "mean_abs_dscore": float(d.mean()), "max_abs_dfused": float(df.max()),
        "frac_requests_page_order_unchanged": same_page / R, "frac_requests_page_set_unchanged": same_set / R,
        "frac_requests_page_order_unchanged_up_to_ties": same_ties / R,
        "mean_page_overlap": float(np.mean(overlap)), "page": page,
        "kendall_tau_full_list_mean": float(np.mean(taus)) if taus else None,
        "kendall_tau_full_list_min": float(np.min(taus)) if taus else None,
        "kendall_tau_page_mean": float(np.mean(taus_page)) if taus_page else None,
        "note": "model score of every candidate of %d requests x %d, bf16 mode minus PG_PREC_F32 (same recalled rows); page = "
                "the first %d of the ItemRankScore order; Kendall tau over the fused scores (%d requests sampled)"
                % (R, K, page, len(taus)),
    }


def multi_output_leg(pa, o, ctx, table, R, K):
    """A two-output model (probs_ctr / probs_cvr on one trunk, PG_MODEL_DNN3_MULTI) against the one-output model of the
    same trunk: the rank stage alone, bf16, R x K random candidate rows.  The reference's fixtures are such models
    (easyrec_response.go:35-70); one gather + one trunk per item whatever the number of outputs."""
    n = table.rows
    rng = np.random.default_rng(5)
    nI = R * K
    cand = rng.integers(0, n, nI).astype(np.uint32)
    offs = (np.arange(R + 1) * K).astype(np.uint32)
    us = o.synth_rows(o.SEED_QUERY, 0, R, 128)
    d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
    d_out = ctx.malloc(nI * 4 * 8)
    out = {}
    w1 = o.Dnn3Weights()
    models = {1: pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w1.w1, w1.b1, w1.w2, w1.b2, w1.w3, w1.b3, 128))}
    for n_out in (2, 4, 8):
        w = o.Dnn3MultiWeights(n_out)
        models[n_out] = pa.RankModel(ctx, pa.MODEL_DNN3_MULTI, pa.PREC_BF16, pa.pack_dnn3_multi(w.w1, w.b1, w.w2, w.b2, w.w3m, w.b3m, 128))
    for n_out, m in models.items():
        best = 1e9
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m.rank_dnn3_dev(table, d_u, d_c, d_o, R, nI, d_out)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        out["outputs_%d_ms" % n_out] = best * 1e3
        m.destroy()
    for n_out in (2, 4, 8):
        out["outputs_%d_vs_1" % n_out] = out["outputs_%d_ms" % n_out] / out["outputs_1_ms"]
    out["note"] = ("rank stage alone, 256-512-256-n_out bf16, %d x %d candidates; k separate one-output models would cost k x outputs_1_ms"
                   % (R, K))
    for p_ in (d_u, d_c, d_o, d_out):
        ctx.free(p_)
    return out


def cfg1_leg(pa, o, ctx):
    """BASELINE.json configs[0]: 1M x 64 in-memory vector table, dot-product top-200, sort.item_score (ascending) —
    the reference's CPU-runnable case.  CPU row = the oracle (scan + heap top-K + sort) on all host cores; the GPU
    row is the same request through the C ABI (host buffers in and out)."""
    n, d, k, R = 1_000_000, 64, 200, 64
    cores = os.cpu_count() or 1
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, R, d)
    t0 = time.time()
    rows, scores = o.recall_topk(tab, q, k, threads=cores)
    for r in range(R):
        o.sort_scores(scores[r].astype(np.float64), False)
    t_cpu = (time.time() - t0) / R
    t0 = time.time()
    o.recall_topk(tab, q[:4], k, threads=1)
    t_cpu1 = (time.time() - t0) / 4
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    g_rows, g_scores, _ = t.recall_topk(q[:1], k)                 # warm-up (builds the shadow)
    lat = []
    for r in range(R):
        t1 = time.perf_counter()
        g_rows, g_scores, _ = t.recall_topk(q[r:r + 1], k)
        order = ctx.sort_scores(g_scores[0].astype(np.float64), descending=False)
        lat.append(time.perf_counter() - t1)
    ok = bool(np.array_equal(g_rows[0], rows[R - 1]) and np.array_equal(order, o.sort_scores(scores[R - 1].astype(np.float64), False)))
    t0 = time.perf_counter()
    t.recall_topk(q, k)                                            # 64 requests in one pass
    t_batch = (time.perf_counter() - t0) / R
    t.destroy()
    return {
        "workload": "configs[0]: 1M x 64 fp32, dot-product top-200, ItemScore (ascending) sort; one request per call",
        "cpu": {"requests_per_s": 1.0 / t_cpu, "ms_per_request": t_cpu * 1e3, "cores": cores, "kind": "port",
                "single_thread_ms_per_request": t_cpu1 * 1e3,
                "bytes_per_request": n * d * 4, "gbs": n * d * 4 / t_cpu / 1e9},
        "gpu": {"requests_per_s": 1.0 / float(np.median(lat)), "p50_ms": float(np.median(lat)) * 1e3,
                "ms_per_request_batched_64": t_batch * 1e3, "matches_oracle": ok},
    }


def l2_recall_leg(pa, o, ctx, rows, K):
    """HologresVectorRecallV2's metric (service/recall/hologres_vector_recall_v2.go:23): top-K by smallest squared Euclidean
    distance on the benchmark's table shape, host buffers in and out — 1 and 128 queries per pass, normalised rows (one
    integer cutoff per 32-row block) and N(0,1) rows (per-row test), with a slice checked against the oracle."""
    d = 128
    t = pa.Table(ctx, rows, d)
    out = {"workload": "squared-Euclidean top-%d of %d x %d fp32 rows (exact; int8 / 4-bit shadows as filters)" % (K, rows, d)}
    rng = np.random.default_rng(17)
    for name in ("normalised_rows", "gaussian_rows"):
        if name == "normalised_rows":
            t.fill_synthetic(o.SEED_TABLE)
        else:
            t.fill_gaussian(o.SEED_TABLE, 1.0)
        res = {}
        for nq in (1, 128):
            q = rng.standard_normal((nq, d)).astype(np.float32) if name == "gaussian_rows" else o.synth_rows(o.SEED_QUERY, 0, nq, d)
            t.recall_topk_l2(q, K)                          # warm-up (builds shadows and row norms)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                t.recall_topk_l2(q, K)
                best = min(best, time.perf_counter() - t0)
            res["ms_per_pass_%d_queries" % nq] = best * 1e3
        res["requests_per_s_at_128"] = 128 / (res["ms_per_pass_128_queries"] * 1e-3)
        out[name] = res
    m = min(rows, 200_000)
    tab = t.download(0, m)
    ts = pa.Table(ctx, m, d)
    ts.upload(tab)
    q = rng.standard_normal((6, d)).astype(np.float32)
    r_, d_, _ = ts.recall_topk_l2(q, 100)
    orow, od = o.recall_topk_l2(tab, q, 100)
    out["slice_matches_oracle"] = bool(np.array_equal(r_, orow) and np.array_equal(d_.view(np.uint32), od.view(np.uint32)))
    ts.destroy()
    t.destroy()
    return out


def where_recall_leg(pa, o, ctx, rows, K):
    """Hologres vector recalls with a WhereClause (hologres_vector_recall.go:23,56-61): `create_time >= constant` over an int32
    item column on the benchmark's table shape, inner product, host buffers in and out — the per-call filter
    (pg_recall_topk_where) and a filtered view (pg_table_view_create) at three selectivities, a slice checked against the
    oracle run on the admitted rows alone."""
    d = 128
    t = pa.Table(ctx, rows, d)
    t.fill_synthetic(o.SEED_TABLE)
    feats = pa.Features(ctx, rows)
    rng = np.random.default_rng(23)
    col = rng.integers(0, 1_000_000, rows).astype(np.int32)
    feats.set_column("create_time", pa.F_I32, col)
    out = {"workload": "inner-product top-%d of the rows of a %d x %d fp32 table that `create_time >= c` admits (exact)" % (K, rows, d)}

    def best_ms(f):
        f()
        f()
        b = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            f()
            b = min(b, time.perf_counter() - t0)
        return b * 1e3
    for frac in (0.5, 0.1, 0.01):
        c = int(1_000_000 * (1 - frac))
        res = {}
        for nq in (1, 128):
            q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
            res["per_call_ms_%d_queries" % nq] = best_ms(lambda: t.recall_topk_where(feats, "create_time", ">=", c, q, K))
        t0 = time.perf_counter()
        v = t.view(feats, "create_time", ">=", c)
        res["view_build_ms"] = (time.perf_counter() - t0) * 1e3
        res["view_rows"] = v.rows
        for nq in (1, 128):
            q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
            for _ in range(4):
                v.recall_topk(q, K)                        # shadows, statistics, threshold model of the view
            res["view_ms_%d_queries" % nq] = best_ms(lambda: v.recall_topk(q, K))
        res["view_requests_per_s_at_128"] = 128 / (res["view_ms_128_queries"] * 1e-3)
        if frac == 0.01:
            q = o.synth_rows(o.SEED_QUERY, 500, 3, d)
            idx = np.nonzero(col >= c)[0]
            full_head = t.download(0, min(rows, 4_000_000))
            keep = idx[idx < full_head.shape[0]]
            orow, osc = o.recall_topk(full_head[keep], q, 50)
            # the same filter restricted to the head of the table: a second column that also excludes the tail
            col2 = col.copy()
            col2[full_head.shape[0]:] = -1
            feats.set_column("create_time_head", pa.F_I32, col2)
            r1, s1, _ = t.recall_topk_where(feats, "create_time_head", ">=", c, q, 50)
            v2 = t.view(feats, "create_time_head", ">=", c)
            r2, s2, _ = v2.recall_topk(q, 50)
            v2.destroy()
            want = keep[orow.astype(np.int64)].astype(np.uint64)
            out["slice_matches_oracle"] = bool(np.array_equal(r1, want) and np.array_equal(r2, want) and
                                               np.array_equal(s1.view(np.uint32), osc.view(np.uint32)) and
                                               np.array_equal(s2.view(np.uint32), osc.view(np.uint32)))
        v.destroy()
        out["admitted_%g" % frac] = res
    feats.destroy()
    t.destroy()
    return out


def cfg4_leg(pa, o, ctx, R, K):
    """BASELINE.json configs[3]: FM (8 + 8 fields, k = 16) + two-tower (128 → 256 → 64) rank of R x K candidates,
    field tables of 1M rows each.  HBM-gather bound: 544 algorithmic bytes per item."""
    vocab = 1_000_000
    fw = o.Fm2tWeights(vocab=vocab)
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
    rng = np.random.default_rng(5)
    users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
    ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
    ifids = rng.integers(0, vocab, (R * K, 8)).astype(np.int32)
    off = (np.arange(R + 1) * K).astype(np.uint32)
    n = R * K
    d_u, d_uf, d_if, d_off = ctx.to_device(users), ctx.to_device(ufids), ctx.to_device(ifids), ctx.to_device(off)
    d_out = ctx.malloc(n * 4)
    from pairec_amd import _lib

    def call():
        _lib.check(ctx.L.pg_rank_fm2t_dev(ctx.h, m.h, d_u, d_uf, d_if, d_off, R, n, d_out))
    for _ in range(3):
        call()
    ctx.synchronize()
    dev_ms = []
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        call()
        dev_ms.append(ctx.stats().last_rank_ms)       # HIP events around the rank launches (synchronises)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = float(np.mean(dev_ms))
    # spot check against the oracle on one request's first items
    got = np.zeros(n, dtype=np.float32)
    ctx.d2h(got, d_out)
    ref = o.fm2t_forward(fw, 1, users[0], ufids[0], ifids[:64])
    err = float(np.max(np.abs(got[:64].astype(np.float64) - ref)))
    per_field_ms = ms
    # the product path: candidates are rows of an item catalogue whose field ids are static columns, the item side is
    # materialised once (pg_fm2t_item_rows_build: one 640-B record per item) and gathered with ONE access per candidate
    n_cat = 20_000_000
    feats = pa.Features(ctx, n_cat)
    cols = ["if%d" % f for f in range(8)]
    for c_ in cols:
        feats.set_column(c_, pa.F_I32, rng.integers(0, vocab, n_cat).astype(np.int32))
    t_b = time.perf_counter()
    ir = pa.ItemRows(m, feats, cols)
    build_s = time.perf_counter() - t_b
    cand = rng.integers(0, n_cat, n).astype(np.uint32)
    d_c = ctx.to_device(cand)

    def call_ir():
        _lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, n, d_out))
    for _ in range(3):
        call_ir()
    ctx.synchronize()
    dev_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        call_ir()
        dev_ms.append(ctx.stats().last_rank_ms)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = float(np.mean(dev_ms))
    ctx.d2h(got, d_out)
    cat_ids = np.stack([feats.gather_i32([c_], cand[:64])[:, 0] for c_ in cols], axis=1)
    ref = o.fm2t_forward(fw, 1, users[0], ufids[0], cat_ids)
    err = float(np.max(np.abs(got[:64].astype(np.float64) - ref)))
    ctx.free(d_c)
    ir.destroy()
    feats.destroy()
    for p in (d_u, d_uf, d_if, d_off, d_out):
        ctx.free(p)
    callers = cfg4_callers_leg(pa, o, ctx, m, fw, users, ufids, R, K)
    m.destroy()
    gbs = n * FM2T_BYTES_PER_ITEM / (ms * 1e-3) / 1e9
    tf = n * FM2T_FLOPS_PER_ITEM / (ms * 1e-3) / 1e12
    return {
        "workload": "configs[3]: FM(8+8 fields, k=16, 1M-row field tables) + two-tower(128-256-64) rank, %d x %d candidates of a "
                    "%d-item catalogue (item side materialised: one 640-B record per item), bf16" % (R, K, n_cat),
        "value": n / (ms * 1e-3), "unit": "ranked items/s", "device_ms_per_step": ms, "wall_ms_per_step": wall * 1e3,
        "max_abs_err_vs_oracle_64_items": err,
        "item_records_build_s": build_s,
        "per_field_path": {"device_ms_per_step": per_field_ms, "value": n / (per_field_ms * 1e-3),
                           "note": "pg_rank_fm2t_dev: 8 ids + 8 scattered 64-B embedding rows per item (1056 B of HBM traffic)"},
        "concurrent_callers": callers,
        "roofline": {"bound": "hbm", "kernel": "pg::fm2t_isw_kernel (FM + item tower over materialised item records: every wave a whole "
                                                "pipeline over 32-item tiles, towers in LDS, X / H1 in registers, records one tile ahead)",
                     "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "bytes_per_item": FM2T_BYTES_PER_ITEM, "traffic": None,
                     "ms_basis": "device_ms_per_step = HIP events around the whole rank stage (tile table + user tower / FM prefix "
                                 "kernel + the rank kernel); the rank kernel alone is kernel_ms_from_profile",
                     # the materialised record is five 128-B lines (544 B used)
                     "hbm_bytes_per_item_moved": 640,
                     "frac_on_bytes_moved": n * 640 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "mfma_achieved_tflops": tf, "mfma_frac": tf / MFMA_BF16_PEAK_TFLOPS},
    }


def cfg4_callers_leg(pa, o, ctx, m, fw, users, ufids, R, K, n_items=4_000_000, callers=768):
    """cfg 4 through per-request plug-in calls: the FM + two-tower IAlgorithm.Run (GpuFm2tAlgorithm) from `callers`
    host threads, one call = one user's candidates (a whole request's 5000, and the reference's BatchCount = 100),
    against one caller alone and against the caller-made batch through host buffers (pg_rank_fm2t_rows, 256 x 5000)."""
    rng = np.random.default_rng(6)
    feats = pa.Features(ctx, n_items)
    cols = ["if%d" % f for f in range(8)]
    for c_ in cols:
        feats.set_column(c_, pa.F_I32, rng.integers(0, fw.vocab, n_items).astype(np.int32))
    table = pa.Table(ctx, 4096, 128)                  # (the scene's table: the FM algorithm only reads the feature columns)
    table.fill_synthetic(o.SEED_TABLE)
    pool = np.ascontiguousarray(rng.integers(0, n_items, 1 << 20).astype(np.uint32))
    users = np.ascontiguousarray(users, dtype=np.float32)
    ufids = np.ascontiguousarray(ufids, dtype=np.int32)
    out = {}
    try:
        # caller-made batch, host buffers in and out (what the coalesced calls can at best equal)
        cand = np.ascontiguousarray(pool[:R * K])
        off = (np.arange(R + 1) * K).astype(np.uint32)
        ir = pa.ItemRows(m, feats, cols)
        ir.rank(users, ufids, cand, off)
        t0 = time.perf_counter()
        for _ in range(5):
            ir.rank(users, ufids, cand, off)
        dt = (time.perf_counter() - t0) / 5
        out["caller_made_batch"] = {"ms_per_batch": dt * 1e3, "value": R * K / dt,
                                    "note": "pg_rank_fm2t_irows on 256 x 5000 candidates, inputs and scores across PCIe (pageable host buffers)"}
        for per_call in (K, 100):
            co = pa.Coalescer(ctx, table, K, algos=[("fm2t", m, ir)], max_rank_items=K, depth=3)
            spec = LoadgenSpec(mode=2, user_vecs=users.ctypes.data, n_users=R, dim=128, k=K, top_n=0,
                               user_field_ids=ufids.ctypes.data, n_user_fields=8, cand_pool=pool.ctypes.data,
                               pool_size=pool.shape[0], rank_items=per_call)
            leg = loadgen(co, spec, callers, 2.5, 2, 1, per_call)
            solo = loadgen(co, spec, 1, 1.0, 3, 1, per_call)
            leg.update({"items_per_call": per_call, "solo_caller": solo, "vs_solo": leg["value"] / solo["value"],
                        "vs_caller_made_batch": leg["value"] / out["caller_made_batch"]["value"]})
            out["calls_of_%d" % per_call] = leg
            co.destroy()
        ir.destroy()
    except RuntimeError as e:
        out["error"] = str(e)
    table.destroy()
    feats.destroy()
    out["unit"] = "ranked items/s"
    return out


def cfg5_leg(pa, o, R, K, prec):
    """BASELINE.json configs[4] as far as one GPU goes: ONE shard of the 8-way 1 B x 128 table (125 M rows: 64 GB of fp32
    rows + 16 GB int8 shadow) through the shard-group API (pg_group_recommend) — recall, merge, owner-computes rank,
    fusion, sort, DPPSort on the first 500 of every sorted list (alpha 1, window 10), page of 100.  The per-GPU work of the
    8-GPU configuration; the exchanges between shards are not in it (one GPU per box on this pool)."""
    rows = 125_000_000
    g = pa.ShardGroup([0])
    g.table_create(rows, 128)
    g.table_fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    g.model_load(pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    ex = pa.Expr(RANK_EXPR)
    out = {}
    for dpp in (500, 0):
        q = make_queries(o, 0, R, 128)
        for _ in range(2):                                                    # warm-up: shadow, buffers, both lanes' scratch
            g.recommend(ex, "gpu_dnn", q, K, 100, dpp_candidates=dpp)
        steps = 6
        qs = [make_queries(o, s_ + 1, R, 128) for s_ in range(steps + 1)]
        t0 = time.perf_counter()
        for s_ in range(steps):
            g.recommend(ex, "gpu_dnn", qs[s_], K, 100, dpp_candidates=dpp)
        dt1 = (time.perf_counter() - t0) / steps
        # two steps in flight (pg_group_recommend_begin / _end: one per lane), as the headline does on one GPU
        t0 = time.perf_counter()
        tk = g.recommend_begin(ex, "gpu_dnn", qs[0], K, 100, dpp_candidates=dpp)
        for s_ in range(steps):
            nxt = g.recommend_begin(ex, "gpu_dnn", qs[s_ + 1], K, 100, dpp_candidates=dpp) if s_ + 1 < steps else None
            g.recommend_end(tk)
            tk = nxt
        dt = (time.perf_counter() - t0) / steps
        out["with_dpp" if dpp else "without_dpp"] = {"ms_per_step": dt * 1e3, "value": R * K / dt, "unit": "ranked items/s",
                                                     "ms_per_step_one_at_a_time": dt1 * 1e3}
    g.destroy()
    out["workload"] = ("configs[4], one of 8 shards: 125M x 128 rows, %d requests x top-%d -> DNN3 rank -> fuse -> sort -> "
                       "DPPSort(500 candidates, page 100, window 10); host buffers in and out, two steps in flight" % (R, K))
    return out


class _TableView:
    """(what roofline_block reads off a table: the shard-group owns the real object)"""

    def __init__(self, L, ctx_h, table_h):
        self.L, self.ctx_h, self.h = L, ctx_h, table_h

    def screen_info(self):
        from pairec_amd import _lib
        eb, sc, rs = C.c_int(), C.c_float(), C.c_float()
        _lib.check(self.L.pg_table_screen_info(self.ctx_h, self.h, C.byref(eb), C.byref(sc), C.byref(rs)))
        return eb.value, sc.value, rs.value


def _checksums(torch, tensors):
    """one wrapping 64-bit position-weighted sum per tensor (bit patterns, not values), on the tensors' device"""
    out = []
    for t in tensors:
        x = t.contiguous()
        x = x.view(torch.int64) if x.element_size() == 8 else x.view(torch.int32).to(torch.int64)
        x = x.reshape(-1)
        out.append((x * torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.int64)).sum())
    return torch.stack(out)


def ranks_agree(torch, coll, world, tensors):
    """Every rank of the sharded step must hold the same rows / fused scores / order / page: their checksums, all-gathered
    on the step's own wire (RCCL, or the host-staged gloo of the dev mode), compared on every rank."""
    mine = _checksums(torch, tensors)
    if world == 1:
        return True
    g = torch.empty(world * mine.numel(), dtype=torch.int64, device=mine.device)
    coll.all_gather_into_tensor(g, mine)
    return bool((g.view(world, -1) == mine[None, :]).all().item())


def oracle_step_pages(o, tab, w, q, k, top_n, dpp_c, alpha, window):
    """The sharded step on ONE table through the oracle (the checker): recall → DNN3 (fp32) → RankScore → ItemRankScore →
    DPPSort.doSort; → per request the page's global rows."""
    rows, rec = o.recall_topk(tab, q, k)
    pages = []
    for r in range(q.shape[0]):
        rk = o.dnn3_forward(w, 0, q[r], tab[rows[r].astype(np.int64)])
        fused = o.widen_f32(rk) * (1 + o.widen_f32(rec[r])) ** 0.1
        order = o.sort_scores(fused, True)
        c = min(k, max(top_n, dpp_c))
        head = order[:c]
        emb = o.l2_normalize_f64(tab[rows[r][head].astype(np.int64)].astype(np.float64))
        page = head[o.dpp_with_window(o.dpp_kernel_matrix(emb, fused[head], alpha), top_n, window)]
        pages.append(rows[r][page])
    return pages


def preflight_ranks(pa, o, torch, dist, coll, rank, world, device, share_gpu, ctl=None):
    """Before anything is timed on N > 1 ranks: the sharded step of configs[4] — recall on every rank's row range, the
    all-gather merge, owner-computes rank, the all-reduced score slab, DPP over reduce-scattered rows — on a SMALL table over
    the run's own wire (RCCL with one rank per GPU; host-staged gloo when the ranks share cuda:0), against the single-table
    oracle ON EVERY RANK, and the ranks against each other.  The first run on real peer devices validates itself
    (service/recall.go:126-150 is the fan-in this replaces).  → (dict, ok) — identical on every rank."""
    from pairec_amd.dist import GpuShardEngine, shard_context, shard_range, sharded_step
    n, d, k, R, top_n, dpp_c = 80_003, 128, 300, 7, 25, 90
    res = {"backend": "gloo (host-staged: ranks share one device)" if share_gpu else "nccl", "ranks": world,
           "workload": "sharded step on a %d x %d table, %d requests, k %d, DPP %d -> page %d; f32 rank model" % (n, d, R, k, dpp_c, top_n)}
    err = ""
    agree = True
    eng = None
    # phase 1, local: table, model, engine.  A rank that fails here must not leave its peers inside the collectives of phase 2
    # (they would sit there until the watchdog fires): every rank reports over the CPU control channel first
    try:
        b, e = shard_range(n, world, rank)
        ctx, _stream = shard_context(torch, pa, device)
        t = pa.Table(ctx, e - b, d, row_offset=b)
        tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
        if rank % 2 == 0:
            t.fill_synthetic(o.SEED_TABLE)
        else:
            t.upload(tab[b:e])
        w = o.Dnn3Weights()
        m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
        ex = pa.Expr(RANK_EXPR)
        eng = GpuShardEngine(torch, ctx, t, m, ex, k, R)
    except Exception as ex_:                                    # noqa: BLE001 — reported, then the run stops
        err = "rank %d (setup): %s: %s" % (rank, type(ex_).__name__, ex_)
    setup_bad = torch.tensor([1 if err else 0], dtype=torch.int32)
    if world > 1:
        torch.cuda.synchronize()
        dist.all_reduce(setup_bad, group=ctl)
    if int(setup_bad.item()) != 0:
        res.update({"pages_equal_oracle_on_every_rank": False, "ranks_agree": False, "ok": False,
                    "error": "the set-up of the preflight failed on %d rank(s): its collective phase was skipped" % int(setup_bad.item())})
        if err:
            res["error_on_this_rank"] = err
            print("[bench] preflight: %s" % err, file=sys.stderr, flush=True)
        return res, False
    try:
        for step, (user0, nq) in enumerate(((77, R), (500, R - 2))):
            q = o.synth_rows(o.SEED_QUERY, user0, nq, d)
            tq = torch.from_numpy(q).to(torch.device("cuda", device))
            torch.cuda.synchronize()
            rows, fused, order, page = sharded_step(eng, coll if world > 1 else None, torch, tq, nq, k, top_n,
                                                    {"candidates": dpp_c, "alpha": 1.0, "window": 10})
            torch.cuda.synchronize()
            with torch.cuda.stream(eng.stream):
                agree = ranks_agree(torch, coll, world, (rows, fused, order, page)) and agree
            rows_n = rows.cpu().numpy().astype(np.uint64)
            page_n = page.cpu().numpy().astype(np.int64)
            want = oracle_step_pages(o, tab, w, q, k, top_n, dpp_c, 1.0, 10)
            for r in range(nq):
                if not np.array_equal(rows_n[r][page_n[r]], want[r]):
                    err = err or "rank %d step %d request %d: the page differs from the oracle's" % (rank, step, r)
        del eng
        ex.free()
        m.destroy()
        t.destroy()
        ctx.close()
    except Exception as ex_:                                    # noqa: BLE001 — reported, then the run stops
        err = err or "rank %d: %s: %s" % (rank, type(ex_).__name__, ex_)
    if not agree:
        err = err or "rank %d: rows / fused / order / page differ between ranks" % rank
    # every rank learns whether ANY rank failed (one int per rank over the control channel)
    bad = torch.tensor([1 if err else 0], dtype=torch.int32)
    if world > 1:
        torch.cuda.synchronize()
        dist.all_reduce(bad, group=ctl)                         # (CPU control channel: no collective kernel on the devices)
    res["pages_equal_oracle_on_every_rank"] = int(bad.item()) == 0
    res["ranks_agree"] = agree
    res["ok"] = int(bad.item()) == 0
    if err:
        res["error_on_this_rank"] = err
        print("[bench] preflight: %s" % err, file=sys.stderr, flush=True)
    return res, res["ok"]


def preflight_group(pa, o, devices):
    """The same step through pg_group_* (one process, peer stores + HIP events between the devices — what a cgo host
    calls) on a small table, against the single-table oracle.  → dict with "ok"."""
    n, d, k, R, top_n, dpp_c = 90_001, 128, 400, 7, 40, 120
    res = {"devices": devices, "workload": "pg_group_recommend on a %d x %d table in %d row-range shards, %d requests, k %d, "
                                           "DPP %d -> page %d" % (n, d, len(devices), R, k, dpp_c, top_n)}
    try:
        tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
        w = o.Dnn3Weights()
        g = pa.ShardGroup(devices)
        g.table_create(n, d)
        g.table_fill_synthetic(o.SEED_TABLE)
        g.model_load(pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
        ex = pa.Expr(RANK_EXPR)
        q = o.synth_rows(o.SEED_QUERY, 21, R, d)
        rows, rec, rnk, fus, cnt = g.recommend(ex, "gpu_dnn", q, k, top_n, dpp_candidates=dpp_c, dpp_alpha=1.0, dpp_window=10)
        want = oracle_step_pages(o, tab, w, q, k, top_n, dpp_c, 1.0, 10)
        ok = all(int(cnt[r]) == top_n and np.array_equal(rows[r], want[r]) for r in range(R))
        g.destroy()
        ex.free()
        res["ok"] = bool(ok)
        if not ok:
            res["error"] = "a page differs from the oracle's"
    except Exception as ex_:                                    # noqa: BLE001
        res["ok"] = False
        res["error"] = "%s: %s" % (type(ex_).__name__, ex_)
    if not res["ok"]:
        print("[bench] group preflight: %s" % res.get("error"), file=sys.stderr, flush=True)
    return res


def shard_sub_leg(pa, o, torch, dist, coll, rank, world, device, args, R, K, prec, blob):
    """configs[4] as a sub-object of the default N > 1 line: the table in row-range shards (args_rows_shard rows per rank),
    the sharded step with DPPSort over RCCL, W untimed + K timed steps, max over ranks; every rank's results checked against
    each other on the wire.  (The headline of that line is the replica mode; `--mode shard` makes this the headline.)"""
    from pairec_amd.dist import GpuShardEngine, shard_context, shard_range, sharded_step
    rows = 125_000_000 if args.rows == 100_000_000 else args.rows
    begin, end = shard_range(rows * world, world, rank)
    ctx, _stream = shard_context(torch, pa, device)
    table = pa.Table(ctx, end - begin, args.dim, row_offset=begin)
    table.fill_synthetic(o.SEED_TABLE)
    table.screen_info()
    model = pa.RankModel(ctx, pa.MODEL_DNN3, prec, blob)
    expr = pa.Expr(RANK_EXPR)
    eng = GpuShardEngine(torch, ctx, table, model, expr, K, R)
    dev = torch.device("cuda", device)
    total = args.warmup + args.steps
    t_qs = [torch.from_numpy(make_queries(o, 7000 + s_, R, args.dim)).to(dev) for s_ in range(total)]
    torch.cuda.synchronize()
    dpp = {"candidates": DPP_CANDIDATES, "alpha": DPP_ALPHA, "window": DPP_WINDOW}

    def sync():
        dist.barrier()
        torch.cuda.synchronize()
    for s_ in range(max(args.warmup, 1)):
        sharded_step(eng, coll, torch, t_qs[s_ % total], R, K, args.page, dpp)
    sync()
    t0 = time.perf_counter()
    for s_ in range(args.warmup, total):
        res = sharded_step(eng, coll, torch, t_qs[s_], R, K, args.page, dpp)
    sync()
    elapsed = time.perf_counter() - t0
    with torch.cuda.stream(eng.stream):
        agree = ranks_agree(torch, coll, world, res)
        fused_sorted = bool((torch.gather(res[1], 1, res[2].long()).diff(dim=1) <= 0).all().item())
    tdev = torch.device("cpu") if coll is not dist else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    out = {"value": R * K * args.steps / elapsed, "unit": "ranked items/s", "ms_per_step": elapsed / args.steps * 1e3,
           "scaling": "weak", "ranks_agree": agree, "lists_sorted": fused_sorted, "ok": bool(agree and fused_sorted),
           # the first exchange carries the best K/G + 6 sqrt(K/G) + 8 entries per request (pairec_amd/dist.py); a shard whose
           # last sent entry lands inside the merged top-K makes the step repeat it with the full lists
           "exchange": dict(getattr(eng, "exchange_stats", None) or {}, unpruned_bytes_per_shard=R * K * 12),
           "config": {"workload": "configs[4]: %d x %d fp32 table in %d row-range shards (%d rows each), %d requests x top-%d -> "
                                  "all_gather merge -> owner-computes DNN3 rank (%s) -> all_reduce scores -> fuse -> sort -> "
                                  "DPPSort(%d candidates, page %d, window %d, reduce_scatter rows + all_gather picks)"
                                  % (rows * world, args.dim, world, rows, R, K, args.prec, DPP_CANDIDATES, args.page, DPP_WINDOW),
                      "parallelism": "one process per GPU, RCCL" if coll is dist else "ranks share one device (dev mode)"}}
    del eng
    expr.free()
    model.destroy()
    table.destroy()
    ctx.close()
    return out


def group_sub_leg(pa, o, devices, args, R, K, prec, blob):
    """configs[4] through pg_group_* (one process, N devices, peer stores) as a sub-object of the default N > 1 line;
    run by rank 0 while the other ranks wait."""
    rows = 125_000_000 if args.rows == 100_000_000 else args.rows
    N = len(devices)
    g = pa.ShardGroup(devices)
    g.table_create(rows * N, args.dim)
    g.table_fill_synthetic(o.SEED_TABLE)
    g.model_load(pa.MODEL_DNN3, prec, blob)
    expr = pa.Expr(RANK_EXPR)
    qs = [make_queries(o, 9000 + s_, R, args.dim) for s_ in range(args.warmup + args.steps + 1)]
    kw = dict(dpp_candidates=DPP_CANDIDATES, dpp_alpha=DPP_ALPHA, dpp_window=DPP_WINDOW)
    for s_ in range(max(args.warmup, 1)):
        g.recommend(expr, "gpu_dnn", qs[s_], K, args.page, **kw)
    t0 = time.perf_counter()
    tk = g.recommend_begin(expr, "gpu_dnn", qs[args.warmup], K, args.page, **kw)
    for s_ in range(args.steps):
        nxt = g.recommend_begin(expr, "gpu_dnn", qs[args.warmup + s_ + 1], K, args.page, **kw) if s_ + 1 < args.steps else None
        page = g.recommend_end(tk)
        tk = nxt
    elapsed = time.perf_counter() - t0
    ok = int(page[4].min()) == args.page and bool(np.all(np.isfinite(page[3])))
    exch = dict(g.exchange_stats(), unpruned_bytes_per_shard=R * K * 12)
    g.destroy()
    expr.free()
    return {"value": R * K * args.steps / elapsed, "unit": "ranked items/s", "ms_per_step": elapsed / args.steps * 1e3,
            "scaling": "weak", "ok": bool(ok), "exchange": exch,
            "config": {"workload": "configs[4]: %d x %d fp32 table in %d row-range shards, %d requests x top-%d -> merge -> "
                                   "owner-computes DNN3 rank (%s) -> fuse -> sort -> DPPSort(%d candidates, page %d); one process, "
                                   "pg_group_recommend_begin / _end, two steps in flight"
                                   % (rows * N, args.dim, N, R, K, args.prec, DPP_CANDIDATES, args.page),
                       "parallelism": "row-range shards x%d in one process: peer stores + HIP events, no collective library" % N}}


def inprocess_main(args, R, K):
    """--mode group / router: ONE process over N devices through the C ABI alone (no torch, no collectives library) — the
    boundary a cgo host uses.  group = configs[4] (row-range shards, peer stores, DPPSort); router = replicas of the
    configs[1]+[2] table behind per-request calls.  Same contract as the other modes: W untimed steps, exactly K timed ones,
    one JSON line."""
    import pairec_amd as pa
    from pairec_amd import _lib
    from oracle import oracle as o       # synthetic-data spec + cpu_baseline leg only
    L = _lib.load()
    have = C.c_int()
    _lib.check(L.pg_device_count(C.byref(have)))
    N = args.gpus
    share = os.environ.get("PG_BENCH_SHARE_GPU") == "1"
    if have.value < N and not share:
        sys.stderr.write("bench.py: --gpus %d --mode %s but this process sees %d GPU(s); refusing to run on fewer devices than "
                         "asked for (PG_BENCH_SHARE_GPU=1: logical shards / replicas on cuda:0, a developer check)\n"
                         % (N, args.mode, have.value))
        raise SystemExit(2)
    devices = list(range(N)) if have.value >= N else [0] * N
    physical = len(set(devices))
    pf = None
    if N > 1 and not args.no_preflight:
        # before anything is timed: the group step on a small table over these devices against the oracle
        pf = preflight_group(pa, o, devices)
        if not pf["ok"]:
            print(json.dumps({"metric": "ranked items/sec, 5k-cand DNN rank", "value": None, "unit": "ranked items/s", "n_gpus": physical,
                              "mode": args.mode, "preflight": pf, "error": "the N > 1 parity preflight failed: nothing was timed"}))
            raise SystemExit(3)
    w = o.Dnn3Weights()
    prec = {"bf16": pa.PREC_BF16, "bf16x3": pa.PREC_BF16X3}.get(args.prec, pa.PREC_F32)
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    expr = pa.Expr(RANK_EXPR)
    total_steps = args.warmup + args.steps
    out_extra = {}
    if args.mode == "group":
        g = pa.ShardGroup(devices)
        g.table_create(args.rows * N, args.dim)
        g.table_fill_synthetic(o.SEED_TABLE)
        g.model_load(pa.MODEL_DNN3, prec, blob)
        qs = [make_queries(o, s_, R, args.dim) for s_ in range(total_steps + 1)]
        dpp_c = DPP_CANDIDATES
        for s_ in range(max(args.warmup, 1)):                       # (the first step builds shadows and both lanes' scratch)
            g.recommend(expr, "gpu_dnn", qs[s_ % len(qs)], K, args.page, dpp_candidates=dpp_c, dpp_alpha=DPP_ALPHA,
                        dpp_window=DPP_WINDOW)
        g.recommend(expr, "gpu_dnn", qs[0], K, args.page, dpp_candidates=dpp_c)
        # per-shard scan stage of an un-overlapped step (HIP events on each shard's launch stream)
        scan = []
        for sh in range(N):
            ms, nbytes = C.c_double(), C.c_uint64()
            _lib.check(L.pg_last_scan_kernel_ms(C.c_void_p(L.pg_group_ctx(g.h, sh)), C.byref(ms), C.byref(nbytes)))
            scan.append(ms.value)
        import gc
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        tk = g.recommend_begin(expr, "gpu_dnn", qs[args.warmup], K, args.page, dpp_candidates=dpp_c, dpp_alpha=DPP_ALPHA,
                               dpp_window=DPP_WINDOW)
        for s_ in range(args.steps):
            nxt = g.recommend_begin(expr, "gpu_dnn", qs[args.warmup + s_ + 1], K, args.page, dpp_candidates=dpp_c,
                                    dpp_alpha=DPP_ALPHA, dpp_window=DPP_WINDOW) if s_ + 1 < args.steps else None
            page = g.recommend_end(tk)
            tk = nxt
        elapsed = time.perf_counter() - t0
        gc.enable()
        assert int(page[4].min()) == args.page
        scan_avg_ms = float(max(scan))
        tv = _TableView(L, C.c_void_p(L.pg_group_ctx(g.h, 0)), C.c_void_p(L.pg_group_table(g.h, 0)))
        rf = roofline_block(tv, R, args, args.rows, scan_avg_ms, None)
        rf["per_shard_scan_ms"] = scan
        g.destroy()
        workload = ("configs[4]: %d x %d fp32 table in %d row-range shards (%d rows each), %d requests x top-%d -> merge -> "
                    "owner-computes DNN3 rank (%s) -> fuse -> sort -> DPPSort(%d candidates, page %d, window %d); one process, "
                    "pg_group_recommend_begin / _end, two steps in flight, host buffers in and out"
                    % (args.rows * N, args.dim, N, args.rows, R, K, args.prec, dpp_c, args.page, DPP_WINDOW))
        parallelism = "table row-range shards x%d in one process: peer stores + HIP events, no collective library" % N
        value = R * K * args.steps / elapsed
    else:
        ctxs, tables, models, cos = [], [], [], []
        for dv in devices:
            cx = pa.Context(dv, None)
            tb = pa.Table(cx, args.rows, args.dim)
            tb.fill_synthetic(o.SEED_TABLE)
            tb.screen_info()
            md = pa.RankModel(cx, pa.MODEL_DNN3, prec, blob)
            ctxs.append(cx)
            tables.append(tb)
            models.append(md)
            cos.append(pa.Coalescer(cx, tb, K, md, expr, "gpu_dnn", max_top_n=args.page, depth=3))
        router = pa.Router(cos)
        users = np.ascontiguousarray(o.synth_rows(o.SEED_QUERY, 0, 1000, args.dim))
        spec = LoadgenSpec(mode=0, user_vecs=users.ctypes.data, n_users=1000, dim=args.dim, k=K, top_n=args.page)
        callers = max(args.callers, 1) * N
        hl = host_lib()
        hl.ph_loadgen_run_target.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(LoadgenSpec), C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_uint64, C.POINTER(LoadgenResult)]

        def run(n_req):
            res = LoadgenResult()
            rc = hl.ph_loadgen_run_target(None, router.h, C.byref(spec), callers, 0, 600.0, n_req, C.byref(res))
            if rc or res.errors:
                raise RuntimeError("load generator failed (rc %d, %d errors)" % (rc, res.errors))
            return res
        run((args.calibrate + max(args.warmup, 1)) * R * N)           # threshold models of every replica + warm-up, untimed
        res = run(args.steps * R * N)
        elapsed = res.seconds
        served = router.served().tolist()
        # the scan stage alone on replica 0, one caller-made batch at a time (per-kernel duration without overlap)
        d_qs = [ctxs[0].to_device(make_queries(o, s_, R, args.dim)) for s_ in range(4)]
        import copy
        a1 = copy.copy(args)
        a1.warmup, a1.steps = 1, 8
        router.destroy()
        for c_ in cos:
            c_.destroy()
        _, _, scan_ms = run_headline(pa, ctxs[0], tables[0], models[0], expr, d_qs, a1, R, K, ctxs[0].synchronize)
        rf = roofline_block(tables[0], R, args, args.rows, float(np.mean(scan_ms)), None, ctxs[0].last_scan_kernel()[1])
        out_extra = {"callers": callers, "requests": int(res.requests), "p50_ms": res.p50_ms, "p99_ms": res.p99_ms,
                     "served_per_replica": served}
        for m_ in models:
            m_.destroy()
        for t_ in tables:
            t_.destroy()
        workload = ("configs[1]+[2] behind per-request calls: %d replicas of the %d x %d table, one coalescer per device behind "
                    "pg_router_recommend; %d host threads with one request (top-%d -> DNN3 %s -> fuse -> sort -> page of %d) "
                    "outstanding each; timed region = exactly steps x %d x %d requests"
                    % (N, args.rows, args.dim, callers, K, args.prec, args.page, R, N))
        parallelism = "request-parallel x%d in one process (least-outstanding router), table replicated per GPU, no exchange" % N
        value = res.requests * K / elapsed
    out = {
        "metric": "ranked items/sec, 5k-cand DNN rank (recall top-5000 -> DNN3 -> fuse -> sort%s)"
                  % (" -> DPPSort" if args.mode == "group" else ""),
        "value": value, "unit": "ranked items/s", "n_gpus": physical, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.prec, "data": "synthetic", "mode": args.mode,
        "config": {"workload": workload, "requests_per_step": R * (N if args.mode == "router" else 1),
                   "candidates_per_request": K, "table_rows_per_gpu": args.rows, "dim": args.dim, "parallelism": parallelism},
        "roofline": rf, "cpu_baseline": None, "device": device_info(),
    }
    out.update(out_extra)
    out["preflight"] = pf
    if physical != N:
        out["ranks"] = N
        out["dev_mode"] = "PG_BENCH_SHARE_GPU=1: %d logical %s on ONE device — a correctness run of the N > 1 code, not a measurement" \
                          % (N, "shards" if args.mode == "group" else "replicas")
    if N == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(o, args, R, K)
    print(json.dumps(out))


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    R, K = args.batch, args.k
    assert 1 <= R <= 256
    if args.mode in ("group", "router"):
        # one process drives all N devices; under a launcher only rank 0 works (the others leave before touching a GPU)
        if rank == 0:
            inprocess_main(args, R, K)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                            # does not return
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

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
        # The control channel: flags and the waits around legs that ONE rank runs over all devices (pg_group_*) go over gloo on
        # the CPU.  An RCCL barrier is an all-reduce kernel that spins on every waiting rank's GPU — beside the persistent
        # one-workgroup-per-CU kernels of the leg being timed.
        ctl = None if share_gpu else dist.new_group(backend="gloo")

        def cpu_barrier():
            torch.cuda.synchronize()
            dist.barrier(group=ctl)

    shard = world > 1 and args.mode == "shard"
    from pairec_amd.dist import shard_range, sharded_step, shard_context, GpuShardEngine, HostStagedCollectives
    # the exchanges of the sharded step: RCCL on device tensors; host-staged gloo when the ranks share cuda:0 (dev mode)
    coll = HostStagedCollectives(dist, torch) if share_gpu else dist
    preflight = None
    group_ok = True
    if world > 1 and not args.no_preflight:
        # the first thing N > 1 ranks do: prove the sharded step on this wire against the oracle (and pg_group_* over the
        # same devices), on every rank; a failure stops the run before a number exists
        preflight, pf_ok = preflight_ranks(pa, o, torch, dist, coll, rank, world, local_rank, share_gpu, ctl)
        # pg_group_* over the same devices (rank 0; the others wait).  The modes of THIS process — replica, shard — do not go
        # through pg_group: a failure there is reported, costs the line its `group` sub-object, and does not stop the run
        # (it is fatal in --mode group / router, whose data path it is).
        flag = torch.zeros(1, dtype=torch.int32)
        if pf_ok and rank == 0:
            preflight["group"] = preflight_group(pa, o, [0] * world if share_gpu else list(range(world)))
            flag += 0 if preflight["group"]["ok"] else 1
        torch.cuda.synchronize()
        dist.all_reduce(flag, group=ctl)                      # (the other ranks wait on the CPU while rank 0 drives their devices)
        group_ok = pf_ok and int(flag.item()) == 0
        preflight["ok"] = pf_ok
        if not pf_ok:
            if rank == 0:
                print(json.dumps({"metric": "ranked items/sec, 5k-cand DNN rank", "value": None, "unit": "ranked items/s",
                                  "n_gpus": 1 if share_gpu else world, "preflight": preflight,
                                  "error": "the N > 1 parity preflight failed: nothing was timed"}))
            cpu_barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
    if shard:
        # one dedicated torch stream shared by the library's kernels and the step's torch ops / RCCL collectives
        ctx, tstream = shard_context(torch, pa, local_rank)
    else:
        ctx = pa.Context(local_rank, None)
    if shard:
        begin, end = shard_range(args.rows * world, world, rank)       # N x rows table, one range per rank
    else:
        begin, end = 0, args.rows                                      # full replica
    table = pa.Table(ctx, end - begin, args.dim, row_offset=begin)

    def fill(dist_name):
        if dist_name == "gaussian":
            table.fill_gaussian(o.SEED_TABLE, 1.0)
        else:
            table.fill_synthetic(o.SEED_TABLE)
    fill(args.table_dist)
    w = o.Dnn3Weights()
    prec = {"bf16": pa.PREC_BF16, "bf16x3": pa.PREC_BF16X3}.get(args.prec, pa.PREC_F32)
    model = pa.RankModel(ctx, pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    expr = pa.Expr(RANK_EXPR)

    total_steps = args.warmup + args.steps
    # replica mode: every rank serves different users
    qs = [make_queries(o, s * (1 if shard else world) + (0 if shard else rank), R, args.dim) for s in range(total_steps)]
    measured_gbs = table.hbm_read_probe(3)           # measured streaming-read ceiling of this GPU's HBM
    table.screen_info()                              # build the shadow outside the timed region

    if not shard:
        d_qs = [ctx.to_device(q) for q in qs]

        extra_ctxs = [pa.Context(local_rank) for _ in range(args.contexts - 1)]

        def sync():
            ctx.synchronize()
            for c_ in extra_ctxs:
                c_.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
        if args.calibrate > 0:
            # service start-up, like the shadow build above: the first batches run on the pilot plan and teach the table's
            # threshold model (other users than the measured ones)
            import copy
            a0 = copy.copy(args)
            a0.warmup, a0.steps = args.calibrate, 0
            d_cal = [ctx.to_device(make_queries(o, 100_000 + s * world + rank, R, args.dim)) for s in range(args.calibrate)]
            run_headline(pa, ctx, table, model, expr, d_cal, a0, R, K, sync, extra_ctxs)
            del d_cal
        predicted0 = sum(c_.stats().recall_predicted for c_ in [ctx] + extra_ctxs)
        rescans0 = sum(c_.stats().recall_rescans for c_ in [ctx] + extra_ctxs)
        pipe, elapsed, scan_ms = run_headline(pa, ctx, table, model, expr, d_qs, args, R, K, sync, extra_ctxs)
        spot = headline_spot_check(o, pipe, table, w, args.prec, qs[(total_steps - 1) % len(qs)], K) if rank == 0 else None
        predicted_batches = sum(c_.stats().recall_predicted for c_ in [ctx] + extra_ctxs) - predicted0
        rescans = sum(c_.stats().recall_rescans for c_ in [ctx] + extra_ctxs) - rescans0
        if extra_ctxs:
            # the roofline figure is a per-kernel property: with batches overlapping on two streams a kernel's event-timed
            # duration includes the other stream's work, so the scan-stage time is measured in a short un-overlapped leg
            # (one context, one batch at a time) right after the headline region
            import copy
            a1 = copy.copy(args)
            a1.warmup, a1.steps = 1, 8
            _, _, scan_ms = run_headline(pa, ctx, table, model, expr, d_qs, a1, R, K, sync)
    else:
        eng = GpuShardEngine(torch, ctx, table, model, expr, K, R)
        dev = torch.device("cuda", local_rank)
        t_qs = [torch.from_numpy(q).to(dev) for q in qs]
        torch.cuda.synchronize()                      # the uploads ran on torch's default stream

        def sync():
            dist.barrier()
            torch.cuda.synchronize()
        # cfg 5: recall + rank + sort.dpp_sort — DPP candidates = top 500 by score, alpha 1, page (ctx.Size) 100, window 10
        dpp = {"candidates": 500, "alpha": 1.0, "window": 10}
        for s in range(args.warmup):
            sharded_step(eng, coll, torch, t_qs[s], R, K, args.page, dpp)
        sync()
        scan_ms = []
        t0 = time.perf_counter()
        for s in range(args.warmup, total_steps):
            last = sharded_step(eng, coll, torch, t_qs[s], R, K, args.page, dpp)
            scan_ms.append(ctx.last_scan_kernel()[0])
        sync()
        elapsed = time.perf_counter() - t0
        # the last timed step's results: identical on every rank (checksums over the wire), every list in ItemRankScore order
        with torch.cuda.stream(eng.stream):
            shard_sanity = {"ranks_agree": ranks_agree(torch, coll, world, last),
                            "lists_sorted": bool((torch.gather(last[1], 1, last[2].long()).diff(dim=1) <= 0).all().item())}
        shard_sanity["ok"] = shard_sanity["ranks_agree"] and shard_sanity["lists_sorted"]
        shard_sanity["exchange"] = dict(getattr(eng, "exchange_stats", None) or {}, unpruned_bytes_per_shard=R * K * 12)
    if os.environ.get("PG_BENCH_STEPTIMES"):          # developer aid
        print("scan ms:", " ".join("%.2f" % x for x in scan_ms), "| rescans", ctx.stats().recall_rescans, file=sys.stderr)
    st = ctx.stats()
    if world > 1:
        dev = torch.device("cpu") if share_gpu else torch.device("cuda", local_rank)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = R * K * args.steps / elapsed * (1 if shard else world)
    scan_avg_ms = float(np.mean(scan_ms))
    rank_items = R * K / (world if shard else 1)
    out = {
        "metric": "ranked items/sec, 5k-cand DNN rank (recall top-5000 of 100M x 128 -> DNN3 -> fuse -> sort)",
        "value": value, "unit": "ranked items/s", "n_gpus": 1 if share_gpu else world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "mode": args.mode if world > 1 else "single",
        "scaling": "weak",
        "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
        "config": {"workload": ("configs[4]: %d x %d fp32 table in %d row-range shards (%d rows each), %d requests x top-%d -> all_gather "
                                "merge -> owner-computes DNN3 rank (%s) -> all_reduce scores -> fuse -> sort -> DPPSort(500 candidates, "
                                "page %d, window 10)" % (args.rows * world, args.dim, world, args.rows, R, K, args.prec, args.page))
                               if shard else
                               ("configs[1]+[2]: recall 5k of %dx%d fp32 table in HBM -> 3-layer DNN rank "
                                "(256-512-256-1, %s) -> RankScore fusion (fp64) -> ItemRankScore sort"
                                % (args.rows, args.dim, {"bf16x3": "bf16 MFMA with split hi + lo operands = bf16x3, fp32-accurate scores",
                                                         "bf16": "bf16 MFMA", "f32": "fp32 MFMA"}[args.prec])),
                   "requests_per_step": R, "candidates_per_request": K, "table_rows": args.rows,
                   "dim": args.dim, "table_dist": args.table_dist, "batches_in_flight": 2 if not shard else 1,
                   "contexts": args.contexts,
                   "calibration_batches": 0 if shard else args.calibrate,
                   "timed_batches_on_predicted_thresholds": None if shard else int(predicted_batches),
                   # the claim "the timed region runs on plan 0" as a checked figure: warm-up + timed batches of the region
                   "batches_in_timed_region_incl_warmup": None if shard else int(args.warmup + args.steps),
                   "all_batches_on_predicted_thresholds": None if shard else bool(predicted_batches >= args.warmup + args.steps),
                   "batches_re_run_after_a_failed_plan": None if shard else int(rescans),
                   "parallelism": ("table row-range shards x%d (%d rows total), all_gather top-K merge + all_reduce scores + DPP top-500"
                                   % (world, args.rows * world)) if shard else
                                  ("request-parallel x%d, table replicated per GPU, no data-path collective" % world
                                   if world > 1 else "1 GPU")},
        "roofline": roofline_block(table, R, args, end - begin, scan_avg_ms, measured_gbs, ctx.last_scan_kernel()[1]),
        "stages_ms": {"recall_device_ms": st.last_recall_ms, "rank_device_ms": st.last_rank_ms},
        "oracle_spot_check": shard_sanity if shard else spot,
        "preflight": preflight,
        "rank_roofline": {"bound": "mfma", "kernel": "pg::dnn3_ws_kernel" if args.prec == "bf16" else "pg::dnn3_x3_kernel<512, 256>",
                          "achieved": rank_items * FLOPS_PER_ITEM / max(st.last_rank_ms, 1e-9) / 1e9,
                          "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": rank_items * FLOPS_PER_ITEM / max(st.last_rank_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS}
        if args.prec != "f32" and st.last_rank_ms > 0 else None,
    }

    if share_gpu:
        out["ranks"] = world
        out["dev_mode"] = ("PG_BENCH_SHARE_GPU=1: %d ranks on ONE device, gloo with host-staged exchanges — a correctness run "
                           "of the N > 1 code, not a measurement" % world)
    solo = world == 1
    if solo and args.latency_reqs > 0:
        # p50 single-request latency (R=1), same pipeline, inputs resident, one batch at a time
        lat = []
        lq = [ctx.to_device(make_queries(o, 7000 + i, 1, args.dim)) for i in range(min(args.latency_reqs, 1000))]
        for i in range(args.latency_reqs):
            dq = lq[i % len(lq)]                      # distinct users
            ctx.synchronize()
            t1 = time.perf_counter()
            pipe.begin(dq, R=1)
            pipe.drain()
            lat.append((time.perf_counter() - t1) * 1e3)
        if os.environ.get("PG_BENCH_LATENCY_TRACE"):
            med = float(np.median(lat))
            print("[bench] single-request latencies above twice the median (index, ms): %s" %
                  [(i, round(v, 2)) for i, v in enumerate(lat) if v > 2 * med], file=sys.stderr, flush=True)
        lat = lat[len(lat) // 10:]
        out["p99_request_latency_ms"] = float(np.percentile(lat, 99))
        out["p50_request_latency_ms"] = float(np.median(lat))
        # the scan stage of that single request (HBM-bound on the shadow it streams: the 4-bit one for dim 128)
        ms1, bytes1 = ctx.last_scan_kernel()
        if ms1 > 0:
            rf1 = roofline_block(table, 1, args, end - begin, ms1, measured_gbs, bytes1)
            out["single_request_roofline"] = {k_: rf1[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                                     "measured_peak", "frac_of_measured", "bytes_per_pass",
                                                                     "ms_per_pass", "traffic_from_profile")}

    if solo and args.prec != "f32" and not args.no_rank_shapes:
        # the rank stage alone, per hidden shape (bf16) and the benchmark's shape in the headline's precision: the rank roofline proper
        for c_ in [ctx] + list(extra_ctxs):
            c_.synchronize()
        shapes = rank_shapes_leg(pa, o, ctx, table, R, K)
        out["rank_shapes"] = shapes
        out["multi_output_rank"] = multi_output_leg(pa, o, ctx, table, R, K)
        bs = [e for e in shapes if e["shape"] == "256-512-256-1" and e.get("prec", "bf16") == args.prec][0]
        out["rank_roofline"] = {"bound": "mfma", "kernel": bs["kernel"], "achieved": bs["achieved"], "peak": bs["peak"],
                                "unit": "TFLOP/s", "frac": bs["frac"], "ms": bs["ms_per_%d_items" % (R * K)],
                                "executed_frac": bs.get("executed_frac"),
                                "measured": "rank stage alone on the device (tile table + request partial + MLP kernel), "
                                            "%d x %d random candidate rows of the resident table; achieved / frac price SURVEY.md 8(d)'s "
                                            "%d flop per item (bf16x3 executes three bf16 products per term: executed_frac)" % (R, K, FLOPS_PER_ITEM),
                                "mfma_busy_from_profile": bs["mfma_busy_from_profile"],
                                "in_pipeline": {"ms": st.last_rank_ms,
                                                "frac": R * K * FLOPS_PER_ITEM / max(st.last_rank_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                                                "note": "HIP events around the stage inside the headline loop, i.e. right behind the scan: the "
                                                        "clock the power cap leaves"}}

    if solo and not args.no_f32_leg:
        # The same timed region in the other precision modes, and what separates each matrix-pipe mode from PG_PREC_F32 (the fp32
        # specification: scores within 2e-7 of the oracle's chains).  north_star's tolerance is 1e-5 on float scores against the
        # reference's fp32 path (algorithm/eas/easyrec_response.go:479-483, eas/tf_response.go:55-59): bf16x3 — the headline —
        # meets it (1.2e-7), plain bf16 does not (4e-5; it passes only against an oracle that rounds where it rounds).
        blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
        precs = {"bf16": pa.PREC_BF16, "bf16x3": pa.PREC_BF16X3, "f32": pa.PREC_F32}
        m32 = model if args.prec == "f32" else pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, blob)
        q_fig = qs[args.warmup % len(qs)]
        if args.prec != "f32":
            out["%s_vs_f32" % args.prec] = precision_figures(pa, ctx, table, expr, model, m32, q_fig, K, args.page)
            out["%s_vs_f32" % args.prec]["note"] = out["%s_vs_f32" % args.prec]["note"].replace("bf16 mode minus", "%s mode (the headline) minus" % args.prec)
        for name in ("bf16", "bf16x3", "f32"):
            if name == args.prec:
                continue
            mm = m32 if name == "f32" else pa.RankModel(ctx, pa.MODEL_DNN3, precs[name], blob)
            pipe_m, el_m, _ = run_headline(pa, ctx, table, mm, expr, d_qs, args, R, K, sync, extra_ctxs)
            ent = {"value": R * K * args.steps / el_m, "unit": "ranked items/s", "ms_per_step": el_m / args.steps * 1e3,
                   "dtype": name, "vs_headline": (R * K * args.steps / el_m) / value,
                   "rank_stage_ms_in_pipeline": ctx.stats().last_rank_ms,
                   "oracle_spot_check": headline_spot_check(o, pipe_m, table, w, name, qs[(total_steps - 1) % len(qs)], K)}
            if name != "f32":
                fig = precision_figures(pa, ctx, table, expr, mm, m32, q_fig, K, args.page)
                fig["note"] = fig["note"].replace("bf16 mode minus", "%s mode minus" % name)
                ent["%s_vs_f32" % name] = fig
                ent["max_abs_dscore_vs_f32_mode"] = fig["max_abs_dscore"]
                ent["frac_requests_page_order_equals_f32_mode"] = fig["frac_requests_page_order_unchanged"]
                ent["frac_requests_page_order_equals_f32_mode_up_to_score_ties"] = fig["frac_requests_page_order_unchanged_up_to_ties"]
            ent["note"] = {"bf16": "the same timed region with plain bf16 operands (configs[2] read literally): faster, and outside "
                                   "north_star's 1e-5 of the fp32 path — its spot check compares with an oracle that mirrors its roundings",
                           "bf16x3": "the same timed region with the rank model at PG_PREC_BF16X3; spot check against the FP32 oracle",
                           "f32": "the same timed region with the rank model at PG_PREC_F32 (fp32 MFMA, scores within 2e-7 of the "
                                  "oracle's fp32 chains); recall, fusion and sort are the same kernels in every mode"}[name]
            if not ent["oracle_spot_check"]["ok"]:
                print("[bench] %s spot check FAILED: %s" % (name, json.dumps(ent["oracle_spot_check"])), file=sys.stderr, flush=True)
            out["%s_mode" % name] = ent
            for b_ in pipe_m.bufs:
                for p_ in b_:
                    ctx.free(p_)
            if name != "f32":
                mm.destroy()
        if m32 is not model:
            m32.destroy()
    if solo and not args.no_batch_sweep:
        out["batch_sweep"] = batch_sweep_leg(pa, o, ctx, table, model, expr, args, K, extra_ctxs, measured_gbs)
    extras = solo and not args.no_extras
    if extras and args.callers > 0:
        out["concurrent_callers"] = concurrent_callers_leg(pa, o, ctx, table, model, expr, args, K)
    if extras and not args.no_power:
        # power and clock under load: the headline loop, the recall pass alone, the rank stage alone (a few seconds each)
        pw = {}
        pp = Pipeline1(pa, ctx, table, model, expr, R, K, extra_ctxs=extra_ctxs)
        it = [0]

        def hstep():
            pp.step(d_qs[it[0] % len(d_qs)])
            it[0] += 1
        pw["headline_loop"] = power_leg(hstep, pp.drain)
        d_rows_, d_sc_ = pp.bufs[0][0], pp.bufs[0][1]

        def rstep():
            table.recall_topk_dev(d_qs[it[0] % len(d_qs)], R, K, d_rows_, d_sc_)
            it[0] += 1
        pw["recall_pass_alone"] = power_leg(rstep, ctx.synchronize)
        rng_ = np.random.default_rng(5)
        cand_ = ctx.to_device(rng_.integers(0, table.rows, R * K).astype(np.uint32))
        offs_ = ctx.to_device((np.arange(R + 1) * K).astype(np.uint32))
        pw["rank_stage_alone"] = power_leg(lambda: model.rank_dnn3_dev(table, d_qs[0], cand_, offs_, R, R * K, pp.bufs[0][2]), ctx.synchronize, sync_each=True)
        ctx.free(cand_)
        ctx.free(offs_)
        for c_ in [ctx] + list(extra_ctxs):
            c_.synchronize()
        for b_ in pp.bufs:
            for p_ in b_:
                ctx.free(p_)
        if any(pw.values()):
            pw["note"] = ("rocm-smi sampled from a thread while the named loop runs back to back for ~3 s (median reading): package power "
                          "against its cap and the shader clock the firmware leaves (maximum 2400 MHz)")
            out["power"] = pw
    if extras:
        # the same headline measurement on the other table distribution (uniform rows are the int8 screen's best case)
        other = "gaussian" if args.table_dist == "uniform" else "uniform"
        fill(other)
        table.screen_info()
        if args.calibrate > 0:
            d_cal = [ctx.to_device(make_queries(o, 100_000 + s * world + rank, R, args.dim)) for s in range(args.calibrate)]
            run_headline(pa, ctx, table, model, expr, d_cal, a0, R, K, sync, extra_ctxs)
            del d_cal
        _, el2, scan2 = run_headline(pa, ctx, table, model, expr, d_qs, args, R, K, sync, extra_ctxs)
        if extra_ctxs:
            _, _, scan2 = run_headline(pa, ctx, table, model, expr, d_qs, a1, R, K, sync)
        rf = roofline_block(table, R, args, end - begin, float(np.mean(scan2)), measured_gbs)
        out["%s_table" % other] = {"table_dist": other, "value": R * K * args.steps / el2, "unit": "ranked items/s",
                                   "ms_per_step": el2 / args.steps * 1e3,
                                   "roofline": {k_: rf[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                                      "frac_survey_8d", "ms_per_pass", "shadow_elem_bytes",
                                                                      "mfma_frac")}}
        if not args.no_clustered:
            out["clustered_table"] = clustered_table_leg(pa, o, ctx, table, model, expr, args, R, K, sync, extra_ctxs, ms_per_step)
        table.destroy()
        out["other_configs"] = {"cfg1": cfg1_leg(pa, o, ctx), "cfg4": cfg4_leg(pa, o, ctx, R, K),
                                "cfg5_one_shard": cfg5_leg(pa, o, R, K, prec)}
        out["l2_recall"] = l2_recall_leg(pa, o, ctx, args.rows, K)
        out["where_recall"] = where_recall_leg(pa, o, ctx, args.rows, K)

    if rank == 0 and extras and not args.no_live_traffic and not os.environ.get("PG_BENCH_CHILD"):
        # roofline.traffic, live: the same scan stage under rocprofv3 --pmc in two child runs (the tables of this process are
        # gone by now; the children build their own)
        tb, detail = measure_traffic_live(args, R)
        out["roofline"]["traffic"] = tb
        out["roofline"]["traffic_detail"] = detail
        if tb:
            out["roofline"]["traffic_over_streamed_bytes"] = tb / out["roofline"]["bytes_per_pass"]
        # ... and cfg 4's rank kernel: HBM bytes per launch of 1.28 M items (scripts/dev/cfg4_prof.py under rocprofv3 --pmc)
        c4 = out.get("other_configs", {}).get("cfg4")
        if c4:
            tb4, det4 = measure_kernel_traffic([os.path.join(ROOT, "scripts", "dev", "cfg4_prof.py"), "random"], "fm2t_isw_kernel")
            c4["roofline"]["traffic"] = tb4
            c4["roofline"]["traffic_detail"] = det4
            if tb4:
                c4["roofline"]["traffic_bytes_per_item"] = tb4 / (R * K)
    if world > 1 and not shard and not args.no_extras:
        # the default N > 1 line carries both §8(e) forms of configs[4] beside the replica headline: the sharded step over
        # RCCL (every rank) and pg_group_* in one process (rank 0, the others wait) — one driver command, both curves
        for c_ in extra_ctxs:
            c_.close()
        model.destroy()
        table.destroy()
        blob5 = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
        try:
            out["shard"] = shard_sub_leg(pa, o, torch, dist, coll, rank, world, 0 if share_gpu else local_rank, args, R, K, prec, blob5)
        except Exception as ex_:                                # noqa: BLE001
            out["shard"] = {"ok": False, "error": "%s: %s" % (type(ex_).__name__, ex_)}
        cpu_barrier()                                         # (ranks 1.. wait on the CPU: nothing of theirs runs beside the group leg)
        if rank == 0 and (preflight is None or group_ok):
            try:
                out["group"] = group_sub_leg(pa, o, [0] * world if share_gpu else list(range(world)), args, R, K, prec, blob5)
            except Exception as ex_:                            # noqa: BLE001
                out["group"] = {"ok": False, "error": "%s: %s" % (type(ex_).__name__, ex_)}
        cpu_barrier()
    failed = False
    if rank == 0:
        out["device"] = device_info()
        if solo and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(o, args, R, K)
        else:
            out["cpu_baseline"] = None
        # a failed oracle check voids the number it belongs to: the diagnostics go to stderr, the value is withheld and the
        # process exits non-zero
        sp = out.get("oracle_spot_check")
        if sp is not None and not sp.get("ok", True):
            print("[bench] oracle spot check FAILED, headline withheld: %s" % json.dumps(sp), file=sys.stderr, flush=True)
            out["value_withheld"] = out["value"]
            out["value"] = None
            failed = True
        for mname in ("bf16_mode", "bf16x3_mode", "f32_mode"):
            mo = out.get(mname)
            if mo and not mo["oracle_spot_check"]["ok"]:
                mo["value_withheld"], mo["value"] = mo["value"], None
                failed = True
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
