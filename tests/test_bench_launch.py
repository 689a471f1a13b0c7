"""bench.py's launcher contract (CPU): `python bench.py --gpus N` with no launcher around it must become N ranks by itself
(the driver invokes it exactly so), report the size of the process group that really ran, and refuse -- loudly -- to
produce an N-GPU line from fewer devices."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_n_without_a_launcher_becomes_n_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["self_launched"] is True and line["dry_launch"] is True
    assert "torch.distributed.run" in r.stderr  # the self-launch announces its command line


def test_gpus_n_with_fewer_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no bench line may be printed"


def test_world_size_must_match_gpus():
    env_args = ["--gpus", "1", "--dry-launch"]
    r = _run(env_args)
    assert r.returncode == 0
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["self_launched"] is False
