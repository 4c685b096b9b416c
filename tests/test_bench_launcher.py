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
