"""cfg 4 (FM + two-tower rank, 256 x 5000 candidates, 1M-row field tables) alone — for rocprofv3 runs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                      # noqa: E402
import pairec_amd as pa           # noqa: E402
from oracle import oracle as o    # noqa: E402

ctx = pa.Context(0)
r = bench.cfg4_leg(pa, o, ctx, 256, 5000)
print(json.dumps({k: v for k, v in r.items() if k != "workload"}))
