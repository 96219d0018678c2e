"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (torch.distributed
backend "nccl") over xGMI.  The reference has no multi-device code; this is the
north_star's "data-parallel across the 8 GPUs with RCCL all-reduce of gradients".

Per step there is exactly ONE collective: a sum all-reduce of the optimizer's flat fp32 gradient bucket
(594 067 floats = 2.4 MB for the default U-Net) issued right after backward; the 1/world_size factor is
folded into the fused Adam kernel (``FusedAdam.grad_scale``).  At this size the all-reduce is
latency-bound, so no bucketing/overlap machinery is needed.  BatchNorm statistics stay
per-rank (no SyncBN in the reference, none added); rank 0 saves checkpoints.
"""
import os

import torch
import torch.distributed as dist


class DataParallelGrads:
    def __init__(self, optimizer, net=None, backend: str = None):
        if not dist.is_initialized():
            raise RuntimeError("call init_distributed() first")
        self.optimizer = optimizer
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.timing = False                                   # bench.py: time every gradient all-reduce
        self._timings = []
        optimizer.prepare()
        optimizer.grad_scale = 1.0 / self.world
        # identical initial weights on every rank (buffers too: BN running stats)
        self.broadcast_state(net)

    def broadcast_state(self, net=None) -> None:
        dist.broadcast(self.optimizer._flat["p"], src=0)
        if net is not None:
            for b in net.buffers():
                dist.broadcast(b, src=0)
        # parameters (views of the flat buffer) and BatchNorm buffers were just rewritten behind their version counters:
        # invalidate the packed-weight images and the cached eval-mode affines / folded heads keyed on them
        from . import engine
        engine.bump_weight_generation()
        engine._bn_generation[0] += 1

    def allreduce_grads(self) -> None:
        opt = self.optimizer
        f = opt._flat
        gbase = f["g"].data_ptr()
        for p, off in zip(f["params"], f["offsets"]):           # gradients normally already live in the bucket
            if p.grad is not None and p.grad.data_ptr() != gbase + 4 * off:
                f["g"][off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = f["g"][off:off + p.numel()].view(p.shape)
        g = f["g"]
        if not self.timing:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return
        # Events on the launch stream around the collective: the first one sits behind backward's last kernel, the
        # second behind the stream-level wait torch inserts for the (R)CCL stream, so the pair holds the all-reduce only.
        if g.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            e1.record()
            self._timings.append((e0, e1))
        else:
            import time
            t0 = time.perf_counter()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            self._timings.append((time.perf_counter() - t0) * 1e3)

    def allreduce_ms(self):
        """Milliseconds of every timed gradient all-reduce so far (``timing = True``); synchronises the device."""
        if any(isinstance(t, tuple) for t in self._timings):
            torch.cuda.synchronize()
        return [t[0].elapsed_time(t[1]) if isinstance(t, tuple) else t for t in self._timings]


def shard_range(n: int, rank: int, world: int):
    """Rank's contiguous, EQUAL-sized range of n samples (the n % world tail is dropped so that every rank runs
    the same number of steps — each step holds one collective)."""
    per = n // world
    if per == 0:
        raise ValueError(f"cannot shard {n} samples over {world} ranks")
    return rank * per, (rank + 1) * per


def shard_train_data(X, y, rank: int, world: int):
    lo, hi = shard_range(len(X), rank, world)
    return X[lo:hi], (None if y is None else y[lo:hi])


def offset_rng_by_rank(rank: int) -> None:
    """Every rank was seeded identically (set_train_rng) so that the nets are drawn identically; from here on the
    random streams that feed DATA (loader shuffles, dropout seeds, the VAE's eps) must differ per rank."""
    if rank:
        import numpy as np
        torch.manual_seed(torch.initial_seed() + rank)           # CPU generator + every GPU generator
        np.random.seed((int(np.random.get_state()[1][0]) + rank) % (2 ** 32))


def init_distributed(backend: str = None, force: bool = False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract); returns (rank, world, local).
    ``force`` creates the process group even for a world of one (the RCCL path on a single GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        # One process per GPU shares the host: torch's intra-op pool defaults to every logical CPU, so 8 ranks would
        # run 8 x 256 spinning OpenMP threads behind each staging copy (the effect measured on the predict pipeline,
        # DESIGN.md §4).  Each rank keeps its share of the cores; AMX_RANK_THREADS overrides.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        share = max(1, (os.cpu_count() or 1) // max(1, local_world))
        torch.set_num_threads(int(os.environ.get("AMX_RANK_THREADS", min(torch.get_num_threads(), share))))
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("AMX_DIST_TIMEOUT_S", "600"))))
    return rank, world, local
