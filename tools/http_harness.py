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


# ---- FeaturesMap (web/features_map.go): the typing of the request's `features` object ------------------------
class _Num(str):
    """A JSON number kept as its literal text (Go's json.Number under decoder.UseNumber())."""


def _parse_number(n: "_Num"):
    # parseNumber (features_map.go:170-183): int64 first, then float64, else the literal
    try:
        i = int(n)
        if "." not in n and "e" not in n.lower() and -2**63 <= i < 2**63:
            return i
    except ValueError:
        pass
    try:
        return float(n)
    except ValueError:
        return str(n)


def _format_value(v) -> str:
    # formatValue (features_map.go:186-195): json.Number → literal text, string → itself, else %v
    if isinstance(v, str):
        return str(v)
    if v is None:
        return "<nil>"
    if isinstance(v, bool):
        return "true" if v else "false"
    return str(v)


def _convert(v):
    """convertValueToString: returns (value, go_type_name)."""
    if v is None:
        return None, "<nil>"
    if isinstance(v, _Num):
        x = _parse_number(v)
        return x, ("int" if isinstance(x, int) else "float64" if isinstance(x, float) else "string")
    if isinstance(v, list):
        if not v:
            return v, "[]interface {}"
        scalar, nested, mixed = 0, 1, 2
        kind = scalar
        for e in v:                                               # convertArray (:55-113)
            if isinstance(e, list):
                if kind == scalar:
                    kind = nested
            elif isinstance(e, str):                              # json.Number or string
                if kind == nested:
                    kind = mixed
            else:
                kind = mixed
            if kind == mixed:
                break
        if kind == nested:
            return [[_format_value(x) for x in e] for e in v if isinstance(e, list)], "[][]string"
        if kind == scalar:
            return [_format_value(e) for e in v], "[]string"
        return v, "[]interface {}"
    if isinstance(v, dict):                                       # convertMap (:116-167)
        has_array = any(isinstance(x, list) for x in v.values())
        has_map = any(isinstance(x, dict) for x in v.values())
        if has_map:
            return {k: _convert(x)[0] for k, x in v.items()}, "map[string]interface {}"
        if has_array:
            out = {}
            for k, x in v.items():
                c, t = _convert(x)
                if t == "[]string":
                    out[k] = c
                elif isinstance(c, list):
                    out[k] = [_format_value(e) for e in c]
                else:
                    out[k] = [_format_value(x)]
            return out, "map[string][]string"
        return {k: _format_value(x) for k, x in v.items()}, "map[string]string"
    if isinstance(v, bool):
        return v, "bool"
    return v, "string"


def features_map(text: str):
    """FeaturesMap.UnmarshalJSON: {name: (value, go type)} — numeric arrays become string arrays so that
    large integers keep every digit (web/features_map_test.go)."""
    temp = json.loads(text, parse_int=_Num, parse_float=_Num)
    if not isinstance(temp, dict):
        raise ValueError("json: cannot unmarshal into FeaturesMap")
    return {k: _convert(v) for k, v in temp.items()}


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
