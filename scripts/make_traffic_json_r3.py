"""profiles/r3_headline_summary.json (scripts/profile_r3_headline.sh: steady state of the 256-request step, thresholds from the
table's model, ONE screened launch per pass) → profiles/r3_scan_traffic.json, the file bench.py's roofline.traffic_from_profile
quotes.  Batch sizes 128 and 1 run kernels that did not change in round 3: their entries are carried over from
profiles/r2_scan_traffic.json."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(ROOT, "profiles", "r3_headline_summary.json")))
old = json.load(open(os.path.join(ROOT, "profiles", "r2_scan_traffic.json")))
pm = s["pmc"]
def find(sub):
    return next(v for k, v in pm.items() if sub in k)
scr, dec, res = find("screen_kernel"), find("screen_decode_kernel"), find("rescore_kernel")
fetch = sum(k["fetch_bytes"] for k in (scr, dec, res))
write = sum(k["write_bytes"] for k in (scr, dec, res))
us = {k: v["last_mean_us"] for k, v in s["kernel_us"].items()}
def us_of(sub):
    return next(v for k, v in us.items() if sub in k)
out = {"_how": s["_how"] + "; per table pass = the pass's three scan-stage launches: screen_kernel over the int8 shadow (leaves hit "
                "records), screen_decode_kernel (records → suspect lists), rescore_kernel (exact fp32 re-scoring gathers); "
                "algorithmic bytes per pass = 12.8e9 (int8 shadow of the 51.2e9-byte fp32 table); entries 128 and 1: profiles/r2_scan_traffic.json",
       "256": {"mfma_busy_frac": round(scr["mfma_busy_frac"], 4),
               "kernels": "screen_kernel + screen_decode_kernel + rescore_kernel (steady state: no pilot sample, no seed scan)",
               "source": "profiles/r3_headline_rocprofv3_summary.txt",
               "screen_kernel_fetch_bytes": int(scr["fetch_bytes"]), "screen_kernel_write_bytes": int(scr["write_bytes"]),
               "screen_decode_kernel_fetch_bytes": int(dec["fetch_bytes"]), "screen_decode_kernel_write_bytes": int(dec["write_bytes"]),
               "rescore_kernel_fetch_bytes": int(res["fetch_bytes"]), "rescore_kernel_write_bytes": int(res["write_bytes"]),
               "screen_kernel_us": round(us_of("screen_kernel"), 1), "screen_decode_kernel_us": round(us_of("screen_decode_kernel"), 1),
               "rescore_kernel_us": round(us_of("rescore_kernel"), 1),
               "valu_per_mfma": round(scr["SQ_INSTS_VALU"] / scr["SQ_INSTS_MFMA"], 2),
               "sustained_clock_ghz": round(scr["GRBM_GUI_ACTIVE"] / 8 / (us_of("screen_kernel") * 1e-6) / 1e9, 2),
               "hbm_bytes_per_pass": int(fetch + write)},
       "128": old["128"], "1": old["1"]}
json.dump(out, open(os.path.join(ROOT, "profiles", "r3_scan_traffic.json"), "w"), indent=1)
print(json.dumps(out["256"], indent=1))
