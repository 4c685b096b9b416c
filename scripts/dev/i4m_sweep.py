"""Developer aid: recall pass time against batch size on the 100 M x 128 table (the sweep bench.py reports as `batch_sweep`),
with the mid-batch 4-bit screen on and off.  GPU box only:  python scripts/dev/i4m_sweep.py [rows]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
t.fill_synthetic(o.SEED_TABLE)
K = 5000
q = o.synth_rows(o.SEED_QUERY, 0, 4096, 128)
d_q = ctx.to_device(q)
d_rows = ctx.malloc(256 * K * 8)
d_sc = ctx.malloc(256 * K * 4)
# train the threshold model (plan 0) as bench.py's calibration does
for i in range(12):
    t.recall_topk_dev(d_q + (i % 8) * 256 * 128 * 4, 256, K, d_rows, d_sc)
if os.environ.get("PG_SWEEP_DEBUG"):
    ctx.set_option("debug_scan", "1")
for mode in os.environ.get("PG_SWEEP_MODES", "i4m,int8").split(","):
    ctx.set_option("no_screen_i4m", "0" if mode == "i4m" else "1")
    for R in [int(x) for x in os.environ.get("PG_SWEEP_R", "1,2,4,8,16,32,48,64,128,256").split(",")]:
        ms, by = [], 0
        for it in range(6):
            off = ((it * 7 + R) % 15) * 256 * 128 * 4
            t0 = time.perf_counter()
            t.recall_topk_dev(d_q + off, R, K, d_rows, d_sc)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            sm, by = ctx.last_scan_kernel()
            if it >= 2:
                ms.append((sm, wall))
        a = np.array(ms)
        print("%-5s R=%3d scan %.3f ms (min %.3f)  wall %.3f ms  bytes %.2f GB  -> %.2f TB/s" %
              (mode, R, a[:, 0].mean(), a[:, 0].min(), a[:, 1].mean(), by / 1e9, by / 1e9 / a[:, 0].mean()), flush=True)
st = ctx.stats()
print("rescans", st.recall_rescans, "predicted", st.recall_predicted)
