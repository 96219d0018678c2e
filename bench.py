#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on its quoted configuration:

    training images/sec, Segmentor U-Net nb_classes=3, 512x512, bs=32 per GPU, fp32 (configs[1]).

A "step" is one full training step of the reference's hot loop (atomai/trainers/trainer.py:189-211):
zero_grad -> forward -> CE loss -> backward -> [RCCL all-reduce of the flat gradient bucket] -> Adam ->
loss.item().  Synthetic data (uniform images, random labels, RandomState(0)), random-init weights
(seed 1), inputs resident in HBM before the timed region.  N>1: one process per GPU (torchrun
contract), weak scaling (bs=32 per GPU), value = images of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline     – the dominant kernel family (MFMA direct convolution: forward + dgrad launches), timed
                 with HIP events on the launch stream over the timed region; algorithmic FLOPs =
                 2*Cin*Cout*k^2*H*W per conv launch (UpsampleBlock 1x1 convs counted at the LOW
                 resolution they are executed at); peak = 157.3 TFLOP/s fp32 MFMA.
  cpu_baseline – oracle/seg_oracle.py (the CPU restatement through stock PyTorch CPU ops) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3     # TFLOP/s, MI355X dense fp32 MFMA (MI355X_MICROARCH.md)
H = W = 512
BS = 32


def unet_conv_table(nb_filters=16, nb_classes=3, hw=512):
    """(name, cin, cout, taps, H) of every conv as EXECUTED (1x1 up-convs at low resolution)."""
    f = nb_filters
    return [("c1.0", 1, f, 9, hw), ("c2.0", f, 2 * f, 9, hw // 2), ("c2.3", 2 * f, 2 * f, 9, hw // 2),
            ("c3.0", 2 * f, 4 * f, 9, hw // 4), ("c3.3", 4 * f, 4 * f, 9, hw // 4),
            ("bn.0", 4 * f, 8 * f, 9, hw // 8), ("bn.3", 8 * f, 8 * f, 9, hw // 8), ("bn.6", 8 * f, 8 * f, 9, hw // 8),
            ("up1", 8 * f, 4 * f, 1, hw // 8), ("c4.0", 8 * f, 4 * f, 9, hw // 4), ("c4.3", 4 * f, 4 * f, 9, hw // 4),
            ("up2", 4 * f, 2 * f, 1, hw // 4), ("c5.0", 4 * f, 2 * f, 9, hw // 2), ("c5.3", 2 * f, 2 * f, 9, hw // 2),
            ("up3", 2 * f, f, 1, hw // 2), ("c6.0", 2 * f, f, 9, hw), ("px", f, nb_classes, 1, hw)]


def step_flops(bs):
    fwd = sum(2.0 * ci * co * t * h * h for _, ci, co, t, h in unet_conv_table())
    first = 2.0 * 1 * 16 * 9 * 512 * 512
    return bs * (3 * fwd - first)           # fwd + dgrad + wgrad, no dgrad for the image itself


class KernelTimer:
    """HIP events (torch.cuda.Event on the current stream == the stream the C-ABI launches on) around
    every call of the selected entry points."""

    def __init__(self, names):
        self.names, self.records, self.active = set(names), [], False
        from atomai_amd import _lib, engine
        self._lib, self._engine, self._orig = _lib, engine, _lib.call

        def call(name, *args):
            if not (self.active and name in self.names):
                return self._orig(name, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig(name, *args)
            e1.record()
            self.records.append((name, args, e0, e1))
            return r
        _lib.call = call

    def summarize(self):
        """-> {name: dict(calls, total_ms, flops)} with algorithmic flops decoded from the call arguments."""
        out = {}
        for name, args, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            fl = 0.0
            if name == "amx_conv2d_fwd":      # (.., C0s@3, .., C1s@7, .., N@16,H@17,W@18,cout@19,taps@20, ..)
                fl = 2.0 * (args[3] + args[7]) * args[19] * args[20] * args[16] * args[17] * args[18]
            elif name == "amx_conv2d_dgrad":  # (dy,aux,k1,k2,k3,bslope,Cs@6,wpk,addend,y,Y0s@10,y1,Y1s@12,..,N@16,H,W,taps@19)
                fl = 2.0 * args[6] * (args[10] + args[12]) * args[19] * args[16] * args[17] * args[18]
                name = "amx_conv2d_fwd"       # same kernel (conv_fwd_kernel): one family
            elif name == "amx_conv2d_wgrad":  # (.., C0s@3, .., C1s@7, dpre@8, Dos@9, part@10, N@11,H,W,cout@14,taps@15)
                fl = 2.0 * (args[3] + args[7]) * args[14] * args[15] * args[11] * args[12] * args[13]
            elif name == "amx_conv2d_wgrad_fused":  # (.., C0s@3, .., C1s@7, dy@8, .., Dos@14, part, bpart, N@17,H,W,cout@20,taps@21)
                fl = 2.0 * (args[3] + args[7]) * args[20] * args[21] * args[17] * args[18] * args[19]
                name = "amx_conv2d_wgrad"
            d = out.setdefault(name, dict(calls=0, total_ms=0.0, flops=0.0))
            d["calls"] += 1
            d["total_ms"] += ms
            d["flops"] += fl
        return out


def cpu_baseline(sample_bs=8, steps=2):
    """Oracle (CPU restatement, stock PyTorch CPU ops) train step on a bounded sample of the workload."""
    from oracle import seg_oracle as so
    torch.set_num_threads(os.cpu_count())
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.rand(sample_bs, 1, H, W).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, 3, (sample_bs, H, W)))
    sd = so.init_unet(3, 16, seed=1)
    opt = so.AdamState(lr=1e-3)
    so.train_step("Unet", sd, opt, x, y, 3)                 # warm-up
    t0 = time.time()
    for _ in range(steps):
        so.train_step("Unet", sd, opt, x, y, 3)
    dt = time.time() - t0
    return {"value": round(sample_bs * steps / dt, 3), "unit": "images/s", "cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{steps} U-Net train steps (fwd+bwd+Adam) at bs={sample_bs}, 512x512, fp32, "
                      f"torch CPU ops, {os.cpu_count()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="run every kernel on one stream (no weight-gradient overlap): the mode of the per-kernel "
                         "HIP-event pass behind `roofline`; used for the rocprofv3 summary that pass must agree with")
    args = ap.parse_args()

    import atomai_amd as aoi
    from atomai_amd.parallel import DataParallelGrads, init_distributed
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    rs = np.random.RandomState(rank)                         # a different shard per rank
    nb = 2                                                   # distinct mini-batches resident per GPU
    X = rs.rand(nb * BS, H, W).astype(np.float32)
    y = rs.randint(0, 3, (nb * BS, H, W))
    model = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
    model.compile_trainer((X, y, X[:BS], y[:BS]), loss="ce", training_cycles=args.steps + args.warmup,
                          batch_size=BS, plot_training_history=False)
    if world > 1:
        model.dp = DataParallelGrads(model.optimizer, model.net)
    timer = None if args.no_kernel_timing else KernelTimer(["amx_conv2d_fwd", "amx_conv2d_dgrad", "amx_conv2d_wgrad", "amx_conv2d_wgrad_fused"])
    from atomai_amd.engine import Tape
    if args.serial:
        Tape.use_side_stream = False

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    losses = []
    wmarks = [time.perf_counter()]
    for i in range(args.warmup):
        losses.append(model.train_step(model.X_train[i % nb], model.y_train[i % nb])[0])
        wmarks.append(time.perf_counter())
    barrier()
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(args.steps):
        losses.append(model.train_step(model.X_train[i % nb], model.y_train[i % nb])[0])
        marks.append(time.perf_counter())                    # loss.item() already synchronised this step
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = np.diff(marks) * 1e3
    ms1 = torch.cuda.memory_stats(dev)
    alloc_info = {"hipMalloc_calls_in_timed_region": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                  "hipFree_calls_in_timed_region": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                  "reserved_GB": round(ms1.get("reserved_bytes.all.current", 0) / 1e9, 2),
                  "peak_allocated_GB": round(ms1.get("allocated_bytes.all.peak", 0) / 1e9, 2)}
    ksteps = 0
    if timer:                      # every rank takes part (the step contains the gradient all-reduce)
        # Per-kernel HIP-event pass, in the same run right after the timed region: the step normally overlaps
        # the weight-gradient kernels with HBM-bound kernels on a second stream, which makes per-launch event
        # durations meaningless, so this pass serialises everything on one stream (it does not enter `value`).
        Tape.use_side_stream = False
        ksteps = min(args.steps, 5)
        timer.active = True
        for i in range(ksteps):
            model.train_step(model.X_train[i % nb], model.y_train[i % nb])
        torch.cuda.synchronize()
        timer.active = False
        Tape.use_side_stream = not args.serial
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    value = world * BS * args.steps / elapsed
    fl = step_flops(BS)
    out = {
        "metric": "training images/sec (512x512, bs=32/GPU) U-Net Segmentor",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Segmentor U-Net nb_classes=3, 512x512, bs=32/GPU, fp32, CE loss, Adam 1e-3 "
                               "(BASELINE.json configs[1]); step = fwd+bwd+allreduce+Adam+loss.item()",
                   "global_batch": world * BS, "parallelism": f"dp{world}",
                   "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)]},
        "step_ms_median_max": [round(float(np.median(per_step)), 3), round(float(per_step.max()), 3)],
        "allocator": alloc_info,
        "step_ms_all": [round(float(v), 1) for v in list(np.diff(wmarks) * 1e3) + list(per_step)],
        "step_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
        "step_frac_of_mfma_f32_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_MFMA_F32, 4),
    }
    if timer:
        summ = timer.summarize()
        conv = summ.get("amx_conv2d_fwd")
        if conv:
            ach = conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12
            traffic = None                   # HBM bytes per launch from the committed rocprofv3 PMC pass
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc))["conv_fwd_family"]["hbm_MB_per_launch"] * 1e6
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32, 4), "traffic": traffic,
                               "kernel": "conv_fwd_kernel<TAPS,NT,HALO> (amx_conv2d_fwd: all forward + dgrad "
                                         "launches of the step)",
                               "launches_per_step": conv["calls"] // ksteps,
                               "ms_per_step": round(conv["total_ms"] / ksteps, 3),
                               "avg_launch_ms": round(conv["total_ms"] / conv["calls"], 4),
                               "measured": f"HIP events on the launch stream, {ksteps} serialised steps after the timed region"}
        wg = summ.get("amx_conv2d_wgrad")
        if wg:
            ach = wg["flops"] / (wg["total_ms"] * 1e-3) / 1e12
            out["roofline_wgrad"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32,
                                     "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32, 4),
                                     "kernel": "wgrad_kernel<TAPS,NT,WM,HALO> (amx_conv2d_wgrad)",
                                     "ms_per_step": round(wg["total_ms"] / ksteps, 3)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
