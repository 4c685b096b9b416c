"""CPU tests of bench.py's N > 1 launch rules (no GPU here): `--gpus N` without a launcher must start its own ranks or
exit non-zero — never print a one-rank line for an N-rank request."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PG_BENCH_SHARE_GPU")}
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=300, env=env)


def test_more_ranks_than_gpus_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing" in r.stderr and '"n_gpus"' not in r.stdout


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4", "--steps", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and '"n_gpus"' not in r.stdout


def test_inprocess_modes_fail_loudly_without_devices():
    import torch
    if torch.cuda.device_count() >= 1:
        import pytest
        pytest.skip("GPU present")
    for mode in ("group", "router"):
        r = _run(["--gpus", "2", "--mode", mode, "--steps", "1"])
        assert r.returncode != 0 and '"n_gpus"' not in r.stdout


def test_launcher_counts_gpus_without_initialising_them():
    """launch_ranks' parent must stay GPU-free (it starts the ranks as children): count_gpus reads the KFD topology or asks
    a short-lived child — it never calls into torch.cuda in this process."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    before = torch.cuda.is_initialized()
    n = bench.count_gpus()
    assert n >= 0
    assert torch.cuda.is_initialized() == before
    # the visibility lists the ranks honour cap the count
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = "0"
        assert bench.count_gpus() <= 1
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        assert bench.count_gpus() == 0
    finally:
        if old is None:
            os.environ.pop("HIP_VISIBLE_DEVICES", None)
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = old


def test_preflight_flag_and_modes_parse():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8", "--no-preflight", "--mode", "shard", "--prec", "bf16x3"]
        a = bench.parse_args()
        assert a.no_preflight and a.rows == 125_000_000 and a.prec == "bf16x3"
        sys.argv = ["bench.py", "--gpus", "8"]
        a = bench.parse_args()
        assert not a.no_preflight and a.mode == "replica" and a.rows == 100_000_000
    finally:
        sys.argv = old


def test_oracle_step_pages_is_the_single_table_pipeline():
    """The preflight's checker: recall -> DNN3 -> RankScore -> sort -> DPP pick sequence on a small table; the page is a
    DPP re-ordering of the head of the sorted list."""
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from oracle import oracle as o
    tab = o.synth_rows(o.SEED_TABLE, 0, 3000, 128)
    q = o.synth_rows(o.SEED_QUERY, 1, 2, 128)
    w = o.Dnn3Weights()
    pages = bench.oracle_step_pages(o, tab, w, q, 100, 10, 30, 1.0, 5)
    rows, rec = o.recall_topk(tab, q, 100)
    for r in range(2):
        assert len(pages[r]) == 10 and len(set(pages[r].tolist())) == 10 and set(pages[r].tolist()) <= set(rows[r].tolist())


def test_power_readings_parse_rocm_smi_text():
    """bench.py's `power` object reads GPU 0's package power, shader clock and cap out of rocm-smi's text (transcribed from an MI355X
    box); anything else — the tool missing, another layout — yields None and the object is left out of the line."""
    sys.path.insert(0, ROOT)
    import bench
    txt = ("\n============================ ROCm System Management Interface ============================\n"
           "================================ Current clock frequencies ================================\n"
           "GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\n"
           "GPU[0]\t\t: sclk clock level: 1: (1535Mhz)\nGPU[0]\t\t: socclk clock level: 0: (28Mhz)\n"
           "=================================== Power Consumption ====================================\n"
           "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1402.0\n"
           "================================== Max Graphics Package Power ==================================\n"
           "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n"
           "================================== End of ROCm SMI Log ===================================\n")
    assert bench.parse_smi(txt) == {"package_w": 1402.0, "sclk_mhz": 1535.0, "cap_w": 1400.0}
    idle = txt.replace("sclk clock level: 1: (1535Mhz)", "sclk clock level: S: (95Mhz)").replace("1402.0", "247.0")
    assert bench.parse_smi(idle) == {"package_w": 247.0, "sclk_mhz": 95.0, "cap_w": 1400.0}
    assert bench.parse_smi(txt.replace("GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n", ""))["cap_w"] is None
    assert bench.parse_smi("rocm-smi: command not found") is None and bench.parse_smi("") is None
