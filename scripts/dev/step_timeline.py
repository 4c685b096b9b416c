"""Developer aid: whole pipeline steps (recall -> rank -> fusion -> sort) of R requests, one batch in flight, for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 scripts/dev/step_timeline.py 32
   python3 scripts/dev/step_timeline.py --parse gpurun_out/tl   (prints the last step's launches: start offset, duration, gap)"""
import sys, os, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    for p in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
        # steps are separated by host round trips: split at gaps > 60 us, print the last three groups
        groups, cur, last_end = [], [], None
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if last_end is not None and s - last_end > 60_000 and cur:
                groups.append(cur)
                cur = []
            cur.append((s, e, r["Kernel_Name"]))
            last_end = max(last_end or 0, e)
        groups.append(cur)
        for g in groups[-4:]:
            t0 = g[0][0]
            print("---- group of %d launches, %.1f us from first start to last end" % (len(g), (max(x[1] for x in g) - t0) / 1e3))
            pe = t0
            for s, e, n in g:
                print("%9.1f %8.1f gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n[:100]))
                pe = e
    sys.exit(0)

import time
import numpy as np
import pairec_amd as pa
from oracle import oracle as o
import bench

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
prec = {"bf16": pa.PREC_BF16, "bf16x3": pa.PREC_BF16X3, "f32": pa.PREC_F32}[sys.argv[3] if len(sys.argv) > 3 else "bf16x3"]
K = 5000
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
model = pa.RankModel(ctx, pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
expr = pa.Expr(bench.RANK_EXPR)
cal = [ctx.to_device(bench.make_queries(o, 100_000 + s, 256, 128)) for s in range(10)]
p256 = bench.Pipeline1(pa, ctx, t, model, expr, 256, K, depth=1)
for d in cal:
    p256.step(d)
p256.drain()
d = [ctx.to_device(bench.make_queries(o, 3000 + 17 * i, R, 128)) for i in range(6)]
one = bench.Pipeline1(pa, ctx, t, model, expr, R, K, depth=1)
for i in range(4):
    one.step(d[i])
one.drain()
ctx.synchronize()
ts = []
for i in range(10):
    t0 = time.perf_counter()
    one.begin(d[i % 6])
    one.drain()
    ts.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
print("wall ms per step:", " ".join("%.3f" % x for x in ts), "| scan", ctx.last_scan_kernel())
