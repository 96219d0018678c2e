"""`gpu` tier: the RCCL branch of the data-parallel path on real silicon.

The builder's lease is one MI355X, so the process group has ONE rank — but it is a real `nccl` group: RCCL is
initialised, the flat CUDA bucket goes through `ncclAllReduce` on RCCL's stream, and the interaction with the tape's
side stream is the one an 8-GPU run has.  The N>1 arithmetic (averaging, shards, schedules) is covered by the gloo
tests (tests/test_data_parallel_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.timeout(900)
def test_one_rank_nccl_group_is_bit_identical_to_no_wrapper():
    r = subprocess.run([sys.executable, os.path.join(HERE, "_nccl_single_rank.py")], env=_env(), capture_output=True,
                       text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["side_stream"]
    assert out["unet_bit_identical"], out["unet_losses"]
    assert out["rvae_bit_identical"], out["rvae_elbos"]
    assert out["fit_dp_attached"] and out["fit_saved"]
    assert out["fit_bit_identical"], out["fit_losses"]


@pytest.mark.timeout(900)
def test_bench_forced_through_the_dp_branch_reports_rccl():
    env = _env()
    env["AMX_BENCH_FORCE_DP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--sustain-seconds", "0", "--no-extra", "--no-cpu-baseline", "--hw", "256", "--bs", "8"],
                       env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["collective_backend"] == "nccl (RCCL)"
    assert out["n_gpus"] == 1 and out["value"] > 0
    lf = out["config"]["loss_first_last"]
    assert lf[1] < lf[0]
