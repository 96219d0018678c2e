"""Deep ensembles across GPUs on CPU: world_size 2 over gloo through the emulated kernels.
Members are independent runs: rank r trains members r, r + 2, ... with no collective on the training path, one
all_gather_object at the end.  Every rank must hold the complete ensemble, BIT-IDENTICAL to a single-process run
(the kernels are deterministic), and only rank 0 writes the metadict."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _data():
    rs = np.random.RandomState(11)
    X = rs.rand(6, 16, 16).astype(np.float32)
    y = rs.randint(0, 3, (6, 16, 16))
    return X, y, X[:2], y[:2]


def _train(tmp, distributed):
    import warnings
    import atomai_amd as aoi
    X, y, Xt, yt = _data()
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(training_cycles=2, batch_size=2, plot_training_history=False,
                                    filename=os.path.join(tmp, "ens_s"))
        _, ens = et.train_ensemble_from_scratch(X, y, Xt, yt, n_models=3, distributed=distributed)
        out["scratch"] = {i: {k: v.cpu().numpy() for k, v in sd.items()} for i, sd in ens.items()}
        et = aoi.trainers.EnsembleTrainer("Unet", nb_classes=3, nb_filters=4)
        et.compile_ensemble_trainer(batch_size=2, plot_training_history=False, filename=os.path.join(tmp, "ens_b"))
        net, ens = et.train_ensemble_from_baseline(X, y, Xt, yt, n_models=2, training_cycles_base=2,
                                                   training_cycles_ensemble=1, distributed=distributed)
        out["baseline"] = {i: {k: v.cpu().numpy() for k, v in sd.items()} for i, sd in ens.items()}
        out["avg"] = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
    return out


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import emu_backend
    emu_backend.use_emulator()
    from atomai_amd.parallel import init_distributed
    init_distributed("gloo")
    mine = os.path.join(tmp, f"rank{rank}")
    os.makedirs(mine, exist_ok=True)
    res = _train(mine, True)
    res["files"] = sorted(os.listdir(mine))
    q.put((rank, res))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(1200)
def test_two_rank_ensemble_equals_single_process(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import emu_backend
    emu_backend.use_emulator()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, res = q.get(timeout=1100)
        got[r] = res
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = os.path.join(str(tmp_path), "single")
    os.makedirs(single)
    ref = _train(single, False)
    for r in (0, 1):
        assert sorted(got[r]["scratch"]) == [0, 1, 2] and sorted(got[r]["baseline"]) == [0, 1]
        for part in ("scratch", "baseline"):
            for i, sd in ref[part].items():
                for k, v in sd.items():
                    assert np.array_equal(got[r][part][i][k], v), (r, part, i, k)
        for k, v in ref["avg"].items():
            assert np.array_equal(got[r]["avg"][k], v), (r, "avg", k)
    assert {"ens_b_ensemble_metadict.tar", "ens_s_ensemble_metadict.tar"} <= set(got[0]["files"])
    assert got[1]["files"] == []                             # only rank 0 writes
    meta = torch.load(os.path.join(str(tmp_path), "rank0", "ens_s_ensemble_metadict.tar"), weights_only=False,
                      map_location="cpu")
    assert sorted(meta["weights"]) == [0, 1, 2]
