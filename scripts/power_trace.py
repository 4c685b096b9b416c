#!/usr/bin/env python3
"""power_trace.py — board power and shader clock WHILE the headline loop runs (evidence for DESIGN.md 4.1a's "the 256-query
screen is power-capped": profiles/r3_screen_clock_ablation.txt derived the clocks from GRBM_GUI_ACTIVE; this samples the
SMU's own telemetry).

For each phase a child `bench.py` runs a long timed region (no extras) and this process samples, every --period seconds,
  power1_average / power1_input (uW), power1_cap (uW), freq1_input (sclk, Hz), temp (if present)
from the GPU's hwmon directory in sysfs (plain file reads: no tool start-up inside the sampling loop); where sysfs has no
such files it falls back to `rocm-smi --showpower --showclocks --json`.  The parent never touches the GPU.

Phases: idle · the headline at 256 requests per pass (stream + int8 MFMA, the benchmark) · at 64 requests per pass
(HBM-bound: a quarter of the matrix work on the same stream) · the headline at PG_PREC_F32 (the rank stage on fp32 MFMA).
Writes a JSON summary (per phase: samples, median / p10 / p90 of power and sclk, the cap, the bench's own items/s) and the
raw samples as CSV next to it.

  python scripts/power_trace.py --out gpurun_out/r4_power            (on the GPU box, via gpurun)
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_hwmon():
    best = None
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        have = [f for f in ("power1_average", "power1_input", "freq1_input") if os.path.exists(os.path.join(d, f))]
        if have and (best is None or len(have) > len(best[1])):
            best = (d, have)
    return best[0] if best else None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def sample_sysfs(d):
    p = read_int(os.path.join(d, "power1_average"))
    if p is None:
        p = read_int(os.path.join(d, "power1_input"))
    return {"power_w": None if p is None else p / 1e6, "sclk_mhz": (read_int(os.path.join(d, "freq1_input")) or 0) / 1e6 or None,
            "mclk_mhz": (read_int(os.path.join(d, "freq2_input")) or 0) / 1e6 or None,
            "temp_c": (read_int(os.path.join(d, "temp1_input")) or 0) / 1e3 or None}


def sample_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        c = d[sorted(d)[0]]
        pw = next((float(v) for k, v in c.items() if "ower" in k and "(W)" in k), None)
        sclk = next((v for k, v in c.items() if k.lower().startswith("sclk")), None)
        mhz = float(str(sclk).split("(")[-1].rstrip("Mhz)").strip()) if sclk else None
        return {"power_w": pw, "sclk_mhz": mhz, "mclk_mhz": None, "temp_c": None}
    except Exception:                                   # noqa: BLE001
        return {"power_w": None, "sclk_mhz": None, "mclk_mhz": None, "temp_c": None}


def tool_snapshot():
    """What the vendor tools say right now (gpu_metrics: current / average socket power, current gfx clocks per XCD) —
    slower to call than sysfs (~1 s), so only every few seconds, from a side thread."""
    out = {}
    for name, cmd in (("amd_smi", ["amd-smi", "metric", "--power", "--clock", "--json"]),
                      ("rocm_smi", ["rocm-smi", "--showpower", "--showclocks", "--json"])):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out[name] = json.loads(r.stdout) if r.stdout.strip().startswith(("{", "[")) else r.stdout[-1500:]
        except Exception as e:                          # noqa: BLE001
            out[name] = "unavailable: %s" % e
    return out


def pct(v, p):
    v = sorted(x for x in v if x is not None)
    return v[min(len(v) - 1, int(p * len(v)))] if v else None


def run_phase(name, bench_args, hw, period, settle, out_rows, snap_after=None):
    env = dict(os.environ)
    proc = None
    if bench_args is not None:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    t0 = time.time()
    samples = []
    snaps = []
    stop = [False]

    def snapper():
        time.sleep(snap_after)
        while not stop[0]:
            sn = tool_snapshot()
            sn["t"] = time.time() - t0
            snaps.append(sn)
            for _ in range(10):
                if stop[0]:
                    break
                time.sleep(0.1)
    import threading
    th = threading.Thread(target=snapper, daemon=True)
    if proc is not None and snap_after is not None:
        th.start()
    while True:
        s_ = sample_sysfs(hw) if hw else sample_smi()
        s_["t"] = time.time() - t0
        samples.append(s_)
        out_rows.append((name, s_["t"], s_["power_w"], s_["sclk_mhz"], s_["mclk_mhz"], s_["temp_c"]))
        if proc is None:
            if s_["t"] > 3.0:
                break
        elif proc.poll() is not None:
            break
        time.sleep(period)
    stop[0] = True
    line = None
    if proc is not None:
        out = proc.stdout.read()
        for ln in out.splitlines():
            if ln.startswith("{"):
                line = json.loads(ln)
    # the timed region sits at the end of the child's life (table fill, shadow build, calibration come first): keep the
    # samples of its last `settle` fraction
    keep = samples if proc is None else samples[int(len(samples) * (1 - settle)):]
    res = {"samples": len(keep), "power_w": {k: pct([x["power_w"] for x in keep], q) for k, q in (("p10", .1), ("median", .5), ("p90", .9))},
           "sclk_mhz": {k: pct([x["sclk_mhz"] for x in keep], q) for k, q in (("p10", .1), ("median", .5), ("p90", .9))},
           "temp_c": pct([x["temp_c"] for x in keep], .5)}
    if snaps:
        # amd-smi's gpu_metrics view is the one that tracks the load (on this box the hwmon files above sit at their idle
        # values whatever runs): socket power and the eight XCDs' gfx clocks per snapshot
        pw, clk = [], []
        for sn in snaps:
            try:
                g0 = sn["amd_smi"]["gpu_data"][0] if isinstance(sn["amd_smi"], dict) else sn["amd_smi"][0]
                pw.append(float(g0["power"]["socket_power"]["value"]))
                cl = [float(v["clk"]["value"]) for k, v in g0["clock"].items() if k.startswith("gfx_") and isinstance(v.get("clk"), dict)]
                if cl:
                    clk.append(sum(cl) / len(cl))
            except Exception:                           # noqa: BLE001
                pass
        keep_n = max(1, int(len(pw) * 0.6))
        res["amd_smi"] = {"snapshots": len(pw), "socket_power_w": pw, "mean_gfx_clk_mhz": clk,
                          "socket_power_w_median_late": pct(pw[-keep_n:], .5), "mean_gfx_clk_mhz_median_late": pct(clk[-keep_n:], .5)}
        res["last_tool_snapshot"] = snaps[-1]
    if line:
        res["bench"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "dtype": line["dtype"],
                        "requests_per_step": line["config"]["requests_per_step"],
                        "scan_ms_per_pass": line["roofline"]["ms_per_pass"], "scan_kernel": line["roofline"]["kernel"]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4_power"))
    ap.add_argument("--period", type=float, default=0.1)
    ap.add_argument("--steps", type=int, default=10000, help="timed steps of the 256-request phase (~4.2 ms each): the SMU's "
                    "telemetry is a slow moving average, a phase has to run for tens of seconds before it settles")
    ap.add_argument("--rows", type=int, default=100_000_000)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    hw = find_hwmon()
    base = ["--rows", str(a.rows), "--warmup", "4", "--no-extras", "--no-cpu-baseline", "--latency-reqs", "0", "--callers", "0",
            "--no-rank-shapes", "--no-f32-leg", "--no-live-traffic"]
    rows = []
    summary = {"_how": "scripts/power_trace.py: hwmon sysfs (%s) sampled every %.0f ms while a child bench.py runs its timed region; "
                       "per phase the last 40 %% of the samples (the timed region); tool_snapshots = amd-smi / rocm-smi read every ~5 s from a side thread" % (hw or "rocm-smi --json fallback", a.period * 1e3),
               "power_cap_w": (read_int(os.path.join(hw, "power1_cap")) or 0) / 1e6 if hw else None,
               "power_cap_max_w": (read_int(os.path.join(hw, "power1_cap_max")) or 0) / 1e6 if hw else None}
    summary["idle"] = run_phase("idle", None, hw, a.period, 1.0, rows)
    summary["headline_256_requests_bf16"] = run_phase("r256", base + ["--steps", str(a.steps)], hw, a.period, 0.4, rows, snap_after=12.0)
    summary["headline_64_requests_bf16"] = run_phase("r64", base + ["--steps", str(a.steps * 3 // 2), "--batch", "64"], hw, a.period, 0.4, rows, snap_after=12.0)
    summary["headline_256_requests_f32_rank"] = run_phase("r256f32", base + ["--steps", str(a.steps // 3), "--prec", "f32"], hw, a.period, 0.4, rows, snap_after=12.0)
    with open(os.path.join(a.out, "power_trace.json"), "w") as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(a.out, "power_trace_samples.csv"), "w") as f:
        f.write("phase,t_s,power_w,sclk_mhz,mclk_mhz,temp_c\n")
        for r in rows:
            f.write(",".join("" if v is None else (v if isinstance(v, str) else "%.4f" % v) for v in r) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
