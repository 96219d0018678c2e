"""bench.py's launch contract, end to end on CPU (gloo + the kernel emulator, tiny shape override):
  * `python bench.py --gpus 2` with NO launcher spawns its two ranks itself and prints ONE JSON line (n_gpus 2);
  * the same command under `python -m torch.distributed.run` (the driver's N>1 command) works too;
  * the line carries the fields the driver's contract names."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--test-backend", "emu", "--hw", "16", "--bs", "2", "--nb-filters", "4", "--steps", "2", "--warmup", "1",
        "--sustain-seconds", "0.01"]
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"]


def _json_lines(stdout: str):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.timeout(600)
def test_plain_command_self_launches_two_ranks():
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY, env=_clean_env(),
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    for k in REQUIRED:
        assert k in out
    assert out["n_gpus"] == 2 and out["config"]["world_size_seen"] == 2 and out["config"]["launcher"] == "self"
    assert out["config"]["global_batch"] == 4 and out["scaling"] == "weak"
    assert out["headline"] is False                       # a shape override is never a headline number
    assert out["sustained"]["steps"] >= 2
    # (`value` is printed with two decimals: on a loaded CPU host it is ~2 img/s, so allow the rounding step)
    assert abs(out["value"] - 2 * 2 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-3 * out["value"] + 0.006


@pytest.mark.timeout(600)
def test_torchrun_launch_two_ranks():
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["launcher"] == "torchrun"


def test_gpus_flag_must_match_world_size():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
