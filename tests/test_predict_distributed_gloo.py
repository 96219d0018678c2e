"""Segmentor predict across GPUs on CPU: world_size 2 over gloo through the emulated kernels.
Each rank decodes its contiguous frame range with the GLOBAL min / ptp (one all-reduce of two floats); rank 0's
gathered output must be BIT-IDENTICAL to the single-process output (eval-mode frames are independent), and so must
the merged Locator coordinates."""
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


def _stack(dtype=np.float32):
    rs = np.random.RandomState(4)
    x = rs.rand(5, 20, 24).astype(dtype)                     # 5 frames over 2 ranks: ranges [0,2) and [2,5)
    x[3] *= 7.5                                              # the global max lives in rank 1's range,
    x[0] -= 2.0                                              # the global min in rank 0's
    return x


def _model():
    import atomai_amd as aoi
    torch.manual_seed(3)
    net, _ = aoi.nets.init_fcnn_model("dilnet", 1, nb_filters=4)
    return net


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    from atomai_amd.parallel import init_distributed
    init_distributed("gloo")
    res = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        p = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=False)
        out = p.run(_stack(dt), compute_coords=False, distributed=True)
        res[name] = out
    p = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=False)
    dec, coords = p.run(_stack(), compute_coords=True, distributed=True, thresh=0.5)
    res["coords"] = (dec, coords)
    lo, part = p.predict(_stack(), distributed=True, gather=False)
    res["nogather"] = (lo, part)
    # more ranks than frames: 1 frame over 2 ranks -> rank 0's range is EMPTY ([0,0)), rank 1 owns the frame
    p = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=True)
    res["one_frame"] = p.run(_stack()[:1], compute_coords=True, distributed=True, thresh=0.5)
    q.put((rank, res))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_predict_is_bit_identical_to_single_process():
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, res = q.get(timeout=800)
        got[r] = res
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        single = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2,
                                             verbose=False).run(_stack(dt), compute_coords=False)
        assert got[0][name].shape == single.shape == (5, 20, 24, 1)
        assert np.array_equal(got[0][name], single), name                  # rank 0: the whole stack, bit for bit
        assert np.array_equal(got[1][name], single[2:5]), name             # other ranks: their own range
    sp = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=False)
    dec1, c1 = sp.run(_stack(), compute_coords=True, thresh=0.5)
    dec, coords = got[0]["coords"]
    assert np.array_equal(dec, dec1) and sorted(coords) == sorted(c1) == list(range(5))
    for i in c1:
        assert np.array_equal(coords[i], c1[i])
    assert got[0]["nogather"][0] == 0 and got[1]["nogather"][0] == 2
    dec0, c0 = sp.run(_stack()[:1], compute_coords=True, thresh=0.5)
    assert got[0]["one_frame"][0].shape == (1, 20, 24, 1) and np.array_equal(got[0]["one_frame"][0], dec0)
    assert list(got[0]["one_frame"][1]) == [0] and np.array_equal(got[0]["one_frame"][1][0], c0[0])
    assert np.array_equal(got[1]["nogather"][1], dec1[2:5])


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "emu"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    from atomai_amd.parallel import init_distributed
    init_distributed("gloo")
    p = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=False)
    dec, coords = p.run(_stack(), compute_coords=True, distributed=True, thresh=0.5)     # 5 frames over 8 ranks
    lo, part = p.predict(_stack(), distributed=True, gather=False)
    q.put((rank, dec if rank == 0 else dec.shape, coords if rank == 0 else None, lo, part.shape))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_rank_predict_with_more_ranks_than_frames():
    """The driver's 8-GPU shape on CPU: 5 frames over 8 ranks — three ranks own an EMPTY frame range, the global
    min / max all-reduce still has 8 participants; rank 0's gathered output is bit-identical to a single process."""
    if torch.cuda.is_available():
        pytest.skip("CPU/gloo tier")
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import emu_backend
    emu_backend.use_emulator()
    import atomai_amd as aoi
    from atomai_amd.predictors import SegPredictor
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(8):
        r, *rest = q.get(timeout=800)
        got[r] = rest
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sp = aoi.predictors.SegPredictor(_model(), use_gpu=False, nb_classes=1, downsampling=2, verbose=False)
    dec1, c1 = sp.run(_stack(), compute_coords=True, thresh=0.5)
    assert np.array_equal(got[0][0], dec1) and sorted(got[0][1]) == list(range(5))
    for i in c1:
        assert np.array_equal(got[0][1][i], c1[i])
    sizes = []
    for r in range(8):
        lo, hi = SegPredictor.frame_range(5, r, 8)
        assert got[r][2] == lo and got[r][3][0] == hi - lo
        sizes.append(hi - lo)
    assert sum(sizes) == 5 and sizes.count(0) == 3


def test_frame_ranges_partition_the_stack():
    from atomai_amd.predictors import SegPredictor
    for n in (0, 1, 5, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            rs = [SegPredictor.frame_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
