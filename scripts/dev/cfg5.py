"""developer aid: cfg 5 leg of bench.py alone (one 125M-row shard through the shard group, one / two steps in flight)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import pairec_amd as pa
from oracle import oracle as o
print(json.dumps(bench.cfg5_leg(pa, o, 256, 5000, pa.PREC_BF16), indent=1))
