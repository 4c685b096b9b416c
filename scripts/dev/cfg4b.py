"""developer aid: cfg 4 leg of bench.py alone (per-field path vs materialised item records, concurrent callers)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import pairec_amd as pa
from oracle import oracle as o
ctx = pa.Context(0)
print(json.dumps(bench.cfg4_leg(pa, o, ctx, 256, 5000), indent=1))
