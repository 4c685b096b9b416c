"""End-to-end p50 / p99 of POST /api/recommend over the GPU engine (tools/http_harness.py) at the benchmark
shape: 100 M x 128 table, recall 5 000, DNN3 rank of all 5 000, ItemRankScore sort, page of 100.
Two scenes over the same engine: "home_feed" runs the stages one plug-in at a time (5 000 Items on the host per
request, as pairec does), "home_feed_page" has ONE recall of Kind "page" that returns the finished page."""
import json
import os
import sys
import threading
import time
import urllib.request

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import http_harness as hh          # noqa: E402
import pairec_amd as pa           # noqa: E402
from oracle import oracle as o    # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
cfg = {
    "RecallConfs": [],
    "SceneConfs": {"home_feed": {"default": {"RecallNames": ["gpu_vector_recall"]}},
                   "home_feed_page": {"default": {"RecallNames": ["gpu_page"]}}},
    "RankConf": {"home_feed": {"RankAlgoList": ["gpu_dnn"], "RankScore": "${gpu_dnn}*(1+${current_score})^0.1",
                               "BatchCount": 5000}},
    "SortNames": {"home_feed": ["ItemRankScore"]},
    "UserDefineConfs": {"pairec_gpu": {"Device": 0,
                                       "Table": {"Rows": rows, "Dim": 128, "IdPrefix": "item_",
                                                 "SyntheticSeed": o.SEED_TABLE},
                                       "Recalls": [{"Name": "gpu_vector_recall", "Kind": "vector", "RecallCount": 5000,
                                                    "RecallAlgo": "gpu_faiss", "ItemType": "video"},
                                                   {"Name": "gpu_page", "Kind": "page", "RecallCount": 5000, "ItemType": "video",
                                                    "RankScore": "${gpu_dnn}*(1+${current_score})^0.1", "RankVar": "gpu_dnn"}],
                                       "Algorithms": [{"Name": "gpu_faiss", "Kind": "faiss"},
                                                      {"Name": "gpu_dnn", "Kind": "dnn3"}]}},
}
h = hh.Harness(cfg)
w = o.Dnn3Weights()
h.load_dnn3(pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
users = o.synth_rows(o.SEED_QUERY, 0, 200, 128)
for i, u in enumerate(users):
    h.set_user_vector("u%d" % i, " ".join("%d:%s" % (k + 1, repr(float(v))) for k, v in enumerate(u)))
srv = hh.make_server(h, 0)
port = srv.server_address[1]
threading.Thread(target=srv.serve_forever, daemon=True).start()


def post(obj):
    req = urllib.request.Request("http://127.0.0.1:%d/api/recommend" % port, data=json.dumps(obj).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=120) as r:
        return json.loads(r.read())


out = {"rows": rows}
pages = {}
for scene in ("home_feed", "home_feed_page"):
    lat = []
    for i in range(220):
        t0 = time.perf_counter()
        r = post({"uid": "u%d" % (i % 200), "size": 100, "scene_id": scene})
        lat.append((time.perf_counter() - t0) * 1e3)
        assert r["code"] == 200 and r["size"] == 100, r.get("msg")
        if i == 0:
            pages[scene] = [x["item_id"] for x in r["items"]]
    lat = lat[20:]
    out[scene] = {"requests": len(lat), "p50_ms": float(np.median(lat)), "p99_ms": float(np.percentile(lat, 99)),
                  "min_ms": float(min(lat))}
out["same_page"] = pages["home_feed"] == pages["home_feed_page"]
print(json.dumps(out))
srv.shutdown()
h.close()
