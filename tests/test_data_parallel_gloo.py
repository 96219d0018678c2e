"""Multi-process data parallelism on CPU (gloo, world_size 2): the N>1 path of bench.py / the trainers.
Each rank trains on its own shard through the emulated kernels; the flat gradient bucket is all-reduced and
the 1/world factor is folded into the fused Adam.  Expected result: the oracle run as 2 replicas on the 2
shards with gradients averaged."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    from atomai_amd.parallel import DataParallelGrads, init_distributed
    init_distributed("gloo")
    rs = np.random.RandomState(10 + rank)
    X = rs.rand(2, 16, 16).astype(np.float32)
    y = rs.randint(0, 3, (2, 16, 16))
    m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4, seed=1 + rank)   # different init per rank ...
    m.compile_trainer((X, y, X, y), training_cycles=2, batch_size=2)
    m.dp = DataParallelGrads(m.optimizer, m.net)                                   # ... broadcast from rank 0
    losses = [m.train_step(m.X_train[0], m.y_train[0])[0] for _ in range(2)]
    q.put((rank, losses, {k: v.detach().cpu().numpy().copy() for k, v in m.net.state_dict().items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_training_matches_averaged_gradient_oracle():
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    from oracle import seg_oracle as so
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, losses, sd = q.get(timeout=500)
        got[r] = (losses, sd)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # oracle: 2 replicas, identical initial weights (rank 0's, seed 1), gradients averaged, one Adam
    sd = so.cast(so.init_unet(3, 4, seed=1), torch.float64)
    data = []
    for r in range(2):
        rs = np.random.RandomState(10 + r)
        data.append((torch.from_numpy(rs.rand(2, 1, 16, 16).astype(np.float32)).double(),
                     torch.from_numpy(rs.randint(0, 3, (2, 16, 16)))))
    opt = so.AdamState(lr=1e-3)
    bn = [OrderedDict(sd), OrderedDict(sd)]
    ref_losses = [[], []]
    for step in range(2):
        grads = []
        for r in range(2):
            rep = OrderedDict((k, (sd[k] if k in so.param_keys(sd) else bn[r][k].clone())) for k in sd)
            loss, _, g = so.loss_and_grads("Unet", rep, data[r][0], data[r][1], 3)
            ref_losses[r].append(float(loss))
            for k in rep:
                if k not in g:
                    bn[r][k] = rep[k]                     # per-rank BatchNorm statistics
            grads.append(g)
        avg = {k: 0.5 * (grads[0][k] + grads[1][k]) for k in grads[0]}
        opt.step(sd, avg)
    for r in range(2):
        np.testing.assert_allclose(got[r][0], ref_losses[r], rtol=1e-4)
    for k in so.param_keys(sd):
        assert np.array_equal(got[0][1][k], got[1][1][k]), k       # replicas stay bit-identical
        a = torch.from_numpy(got[0][1][k]).double()
        assert float((a - sd[k]).abs().max()) < 2e-3 * max(1.0, float(sd[k].abs().max())), k
