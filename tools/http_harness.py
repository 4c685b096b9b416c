#!/usr/bin/env python3
"""`/api/recommend` over the GPU engine: an in-process HTTP harness for end-to-end p50.

Speaks the reference's wire format (web/recommend_controller.go:24-157, web/response.go:3-7):
  request  POST /api/recommend  {"uid": str, "size": int, "scene_id": str, "category": str, "debug": bool,
                                 "features": {...}}
  response {"code": 200|299|400, "msg": str, "request_id": str, "size": n,
            "items": [{"item_id", "item_type", "retrieve_id"}]}
with the controller's checks: empty body → 400 "request body empty"; bad JSON → 400 with the parse error;
missing uid → 400 "uid not empty"; size <= 0 → 10; scene_id "" → "default_scene"; fewer items than `size`
→ code 299 "items size not enough".  The request runs through libpairec_host.so (recall → UniqueFilter →
rank → sort) exactly as tests/test_host_mirror.py drives it; an AB experiment can be attached with the
non-reference field "experiment_params" (the layer-params object).  This is a measurement harness, not a
server: single-threaded on purpose (one request owns the GPU context at a time).

    python tools/http_harness.py --config recconf.json --port 8000 [--dnn3-blob model.bin --prec f32]
"""
import argparse
import ctypes as C
import json
import os
import uuid
from http.server import BaseHTTPRequestHandler, HTTPServer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_host():
    L = C.CDLL(os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
    L.ph_last_error.restype = C.c_char_p
    L.ph_engine_create.restype = C.c_void_p
    L.ph_engine_create.argtypes = [C.c_char_p]
    L.ph_engine_destroy.argtypes = [C.c_void_p]
    L.ph_engine_load_dnn3.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ph_set_user_vector.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.ph_recommend.restype = C.c_char_p
    L.ph_recommend.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
    L.ph_recommend_ab.restype = C.c_char_p
    L.ph_recommend_ab.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
    return L


class Harness:
    def __init__(self, config: dict):
        self.L = load_host()
        self.h = self.L.ph_engine_create(json.dumps(config).encode())
        if not self.h:
            raise RuntimeError("engine: %s" % self.L.ph_last_error().decode())

    def close(self):
        if self.h:
            self.L.ph_engine_destroy(self.h)
            self.h = None

    def set_user_vector(self, uid: str, vec_text: str):
        self.L.ph_set_user_vector(self.h, uid.encode(), vec_text.encode())

    def load_dnn3(self, prec: int, blob: bytes):
        if self.L.ph_engine_load_dnn3(self.h, prec, blob, len(blob)) != 0:
            raise RuntimeError("model: %s" % self.L.ph_last_error().decode())

    # RecommendController.Process + CheckParameter + doProcess
    def handle(self, body: bytes) -> dict:
        request_id = str(uuid.uuid4())

        def error(msg):
            return {"code": 400, "msg": msg, "request_id": request_id}

        if not body:
            return error("request body empty")
        try:
            p = json.loads(body)
            if not isinstance(p, dict):
                raise ValueError("json: cannot unmarshal into RecommendParam")
        except ValueError as e:
            return error(str(e))
        uid = p.get("uid") or ""
        if not isinstance(uid, str) or len(uid) == 0:
            return error("uid not empty")
        size = p.get("size") if isinstance(p.get("size"), int) else 0
        if size <= 0:
            size = 10
        scene = p.get("scene_id") or "default_scene"
        feats = p.get("features")
        if isinstance(feats, dict) and isinstance(feats.get("user_vector"), str):
            self.set_user_vector(uid, feats["user_vector"])          # harness convenience: inline user embedding
        exp = p.get("experiment_params")
        if isinstance(exp, dict):
            out = self.L.ph_recommend_ab(self.h, uid.encode(), size, scene.encode(), json.dumps(exp).encode())
        else:
            out = self.L.ph_recommend(self.h, uid.encode(), size, scene.encode())
        if out is None:
            return {"code": 500, "msg": self.L.ph_last_error().decode(), "request_id": request_id}
        items = [{"item_id": x["item_id"], "item_type": x.get("item_type", ""), "retrieve_id": x["retrieve_id"]}
                 for x in json.loads(out)["items"]]
        if len(items) < size:
            return {"code": 299, "msg": "items size not enough", "request_id": request_id, "size": len(items),
                    "items": items}
        return {"code": 200, "msg": "success", "request_id": request_id, "size": len(items), "items": items}


def make_server(harness: Harness, port: int) -> HTTPServer:
    class Handler(BaseHTTPRequestHandler):
        def do_POST(self):
            if self.path.split("?")[0] != "/api/recommend":
                self.send_error(404)
                return
            n = int(self.headers.get("Content-Length") or 0)
            resp = json.dumps(harness.handle(self.rfile.read(n) if n else b"")).encode()
            self.send_response(200)
            self.send_header("Content-Type", "application/json")
            self.send_header("Content-Length", str(len(resp)))
            self.end_headers()
            self.wfile.write(resp)

        def log_message(self, *a):
            pass

    return HTTPServer(("127.0.0.1", port), Handler)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, help="recconf JSON (RecallConfs / RankConf / SortNames / UserDefineConfs.pairec_gpu)")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--dnn3-blob", help="pg_model_load blob of the DNN3 rank model")
    ap.add_argument("--prec", choices=["f32", "bf16"], default="bf16")
    args = ap.parse_args()
    with open(args.config) as f:
        h = Harness(json.load(f))
    if args.dnn3_blob:
        with open(args.dnn3_blob, "rb") as f:
            h.load_dnn3(0 if args.prec == "f32" else 1, f.read())
    srv = make_server(h, args.port)
    try:
        srv.serve_forever()
    finally:
        h.close()


if __name__ == "__main__":
    main()
