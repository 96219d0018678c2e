"""bench.py's launch contract, end to end on CPU (bench.main() driven by tests/emu/bench_emu.py: gloo + the kernel
emulator, tiny shape override):
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
BENCH = os.path.join(ROOT, "tests", "emu", "bench_emu.py")     # bench.main() on the CPU emulator + gloo (test helper)
TINY = ["--hw", "16", "--bs", "2", "--nb-filters", "4", "--steps", "2", "--warmup", "1",
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
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + TINY, env=_clean_env(),
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


@pytest.mark.timeout(900)
def test_plain_command_self_launches_eight_ranks():
    """The driver's widest shape (`--gpus 8`): eight ranks, one JSON line, and the diagnosis fields a first real 8-GPU
    run will be read by — every rank's own step time and the time of its gradient all-reduces."""
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"] + TINY, env=_clean_env(),
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 8 and out["config"]["world_size_seen"] == 8 and out["config"]["global_batch"] == 16
    assert out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak"
    assert len(out["per_rank_ms_per_step"]) == 8 and all(v > 0 for v in out["per_rank_ms_per_step"])
    assert len(out["allreduce_ms_per_rank_mean"]) == 8 and out["allreduce_ms"] > 0
    assert out["allreduce_bytes"] > 0
    # whole-job value = 8 ranks x bs 2 x steps / the slowest rank's region
    assert abs(out["value"] - 8 * 2 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-3 * out["value"] + 0.006
    assert max(out["per_rank_ms_per_step"]) <= out["ms_per_step"] * 1.001 + 1e-3
    # first-contact fields (how to read them: DESIGN.md section 4 "First 8-GPU run")
    assert set(out["allreduce_ms_percentiles"]) == {"p50", "p90", "p99", "max"}
    assert out["allreduce_ms_percentiles"]["p50"] <= out["allreduce_ms_percentiles"]["max"]
    assert len(out["host_enqueue_ms_per_step_per_rank"]) == 8 and all(v > 0 for v in out["host_enqueue_ms_per_step_per_rank"])
    assert "xgmi_seen" in out and "sclk_mhz_per_rank_min_median_max" in out and out["per_rank_spread_rel"] >= 0


@pytest.mark.timeout(600)
def test_torchrun_launch_two_ranks():
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2"] + TINY
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["launcher"] == "torchrun"


def test_gpus_flag_must_match_world_size():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY, env=env,     # the product script
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.timeout(600)
def test_a_failing_rank_ends_the_job_at_once():
    """when one rank dies the launcher must not leave the others inside the next collective until the
    process-group timeout (minutes).  `--fail-rank 1` makes rank 1 raise after the warm-up; the launcher returns
    non-zero within seconds of that, names the rank and relays ITS stderr tail."""
    import time
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    env = dict(_clean_env(), AMX_DIST_TIMEOUT_S="600")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--fail-rank", "1"] + TINY, env=env,
                       capture_output=True, text=True, timeout=500)
    took = time.time() - t0
    assert r.returncode != 0
    assert "rank 1 of 2 exited with code" in r.stderr, r.stderr[-2000:]
    assert "injected failure" in r.stderr and "[rank 1/2]" in r.stderr, r.stderr[-2000:]
    assert not _json_lines(r.stdout)                      # no measurement line from a broken job
    # start-up (imports, emulator load, 1 warm-up step) dominates; the point is: far below the 600 s collective timeout
    assert took < 120, took


def test_more_gpus_asked_than_visible_is_one_clear_line():
    """`bench.py --gpus N` with fewer than N visible devices: one line, before any rank is spawned (here: no GPU at all)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("needs a host with fewer than 8 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + TINY, env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    msg = r.stderr + r.stdout
    assert r.returncode != 0
    assert ("only" in msg and "visible" in msg) or "needs an MI355X" in msg, msg[-1000:]
    assert "Traceback" not in msg
