"""Turn the rocprofv3 output of scripts/profile.sh (gpurun_out/prof_<tag>_<R>/) into the committed evidence:
profiles/<tag>_<R>_kernel_stats.csv, profiles/<tag>_<R>_rocprofv3_summary.txt and profiles/r2_scan_traffic.json
(the `traffic_from_profile` field of bench.py's roofline object: HBM bytes per pass and the MFMA pipe's busy fraction).

    python scripts/make_traffic_json.py r2a 256 128 1
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN_KERNELS = ("screen_kernel", "screen4_kernel", "rescore_kernel", "scan_kernel")
PASSES = 7  # bench.py --steps 5 --warmup 2 --no-extras --contexts 1
SIMDS = 1024  # 256 CUs x 4


def counter_totals(d, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for p in glob.glob(os.path.join(d, "pmc_%s" % counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] != counter:
                continue
            for key in SCAN_KERNELS:
                if key in row["Kernel_Name"]:
                    tot[key] += float(row["Counter_Value"])
                    n[key] += 1
    return tot, n


def main():
    tag, sizes = sys.argv[1], [int(x) for x in sys.argv[2:]]
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ passes of `python3 bench.py --steps 5 --warmup 2 "
                   "--no-extras --contexts 1 --batch R` (scripts/profile.sh, scripts/make_traffic_json.py); mfma_busy_frac = "
                   "SQ_VALU_MFMA_BUSY_CYCLES of the full-pass screen_kernel launch / (1024 SIMDs x GRBM_GUI_ACTIVE of that "
                   "launch / 8 XCDs); FETCH_SIZE is KB and is doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide 16 B/lane reads); per table pass = sum "
                   "over the pass's scan-stage launches (exact seed of the pilot, screened sample launch, screened "
                   "full pass over the int8 shadow, exact fp32 re-scoring gathers) / 7 passes; algorithmic bytes per "
                   "pass = 12.8e9 (int8 shadow of the 51.2e9-byte fp32 table); batches of <= 4 queries stream the 4-bit shadow "
                   "(7.2e9 bytes) in the full pass, the int8 one in the pilot sample"}
    for R in sizes:
        d = os.path.join(ROOT, "gpurun_out", "prof_%s_%d" % (tag, R))
        f, nf = counter_totals(d, "FETCH_SIZE")
        w, _ = counter_totals(d, "WRITE_SIZE")
        fetch_kb, write_kb = sum(f.values()) / PASSES, sum(w.values()) / PASSES
        # MFMA pipe busy fraction of the dominant launch (the full pass = the screen_kernel dispatch with the most cycles)
        busy = None
        try:
            rows = []
            for pth in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
                rows += [r_ for r_ in csv.DictReader(open(pth)) if "screen_kernel" in r_["Kernel_Name"]]
            mf = [float(r_["Counter_Value"]) for r_ in rows if r_["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
            ga = [float(r_["Counter_Value"]) for r_ in rows if r_["Counter_Name"] == "GRBM_GUI_ACTIVE"]
            if mf and ga:
                busy = max(mf) / (SIMDS * max(ga) / 8.0)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        except Exception:
            busy = None
        out[str(R)] = {
            "mfma_busy_frac": None if busy is None else round(busy, 4),
            "kernels": "screen_kernel (+ screen4_kernel: the 4-bit full pass of batches of <= 4) + rescore_kernel + seed scan_kernel",
            "source": "profiles/%s_%d_rocprofv3_summary.txt" % (tag, R),
            "fetch_kb_per_pass": round(fetch_kb), "write_kb_per_pass": round(write_kb),
            "screen_kernel_fetch_kb_per_pass": round(f["screen_kernel"] / PASSES),
            "screen4_kernel_fetch_kb_per_pass": round(f["screen4_kernel"] / PASSES),
            "rescore_kernel_fetch_kb_per_pass": round(f["rescore_kernel"] / PASSES),
            "dispatches": dict(nf),
            "hbm_bytes_per_pass": round((2 * fetch_kb + write_kb) * 1024),
        }
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", "%s_%d_rocprofv3_summary.txt" % (tag, R)))
        ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
        shutil.copy(ks, os.path.join(ROOT, "profiles", "%s_%d_kernel_stats.csv" % (tag, R)))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r2_scan_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_pass"] for k, v in out.items() if k != "_how"}))


if __name__ == "__main__":
    main()
