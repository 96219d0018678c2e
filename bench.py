#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on its quoted configuration:

    training images/sec, Segmentor U-Net nb_classes=3, 512x512, bs=32 per GPU, fp32 (configs[1]).

A "step" is one full training step of the reference's hot loop (atomai/trainers/trainer.py:189-211):
zero_grad -> forward -> CE loss -> backward -> [RCCL all-reduce of the flat gradient bucket] -> Adam ->
loss.item().  Synthetic data (uniform images, random labels, RandomState(rank)), random-init weights
(seed 1), inputs resident in HBM before the timed region.

Launching.  `python bench.py --gpus N` works both ways:
  * under a launcher (torch.distributed.run / torchrun): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read
    from the environment, one rank per GPU;
  * plain (no WORLD_SIZE in the environment) with N > 1: this process spawns the N ranks itself
    (127.0.0.1 rendezvous on a free port) and relays rank 0's JSON line.
Weak scaling (bs=32 per GPU); value = images of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      – the dominant kernel family (MFMA direct convolution: forward + dgrad launches), timed
                  with HIP events on the launch stream; algorithmic FLOPs = 2*Cin*Cout*k^2*H*W per conv
                  launch (UpsampleBlock 1x1 convs counted at the LOW resolution they are executed at,
                  SURVEY.md §8-d); peak = 157.3 TFLOP/s fp32 MFMA.
  sustained     – the same step repeated for >= --sustain-seconds after the timed region (steady-state
                  clocks): images/s, min/median/max step, shader clock samples when sysfs exposes them.
  extra_configs – bounded one-liners for BASELINE.json configs[2..4] (dilnet predict, rVAE step, DKL
                  covariance) timed in the same process (N=1 only).
  cpu_baseline  – oracle/seg_oracle.py (the CPU restatement through stock PyTorch CPU ops) timed on the
                  host cores at the best thread count of a small sweep (rank 0, N=1 only).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3     # TFLOP/s, MI355X dense fp32 MFMA (MI355X_MICROARCH.md)
H = W = 512
BS = 32
_pmc = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_hbm_traffic.json")))
PMC_FILE = os.path.join("profiles", os.path.basename(_pmc[-1])) if _pmc else "profiles/none"    # the latest round's PMC pass


def unet_conv_table(nb_filters=16, nb_classes=3, hw=512):
    """(name, cin, cout, taps, H) of every conv as EXECUTED (1x1 up-convs at low resolution)."""
    f = nb_filters
    return [("c1.0", 1, f, 9, hw), ("c2.0", f, 2 * f, 9, hw // 2), ("c2.3", 2 * f, 2 * f, 9, hw // 2),
            ("c3.0", 2 * f, 4 * f, 9, hw // 4), ("c3.3", 4 * f, 4 * f, 9, hw // 4),
            ("bn.0", 4 * f, 8 * f, 9, hw // 8), ("bn.3", 8 * f, 8 * f, 9, hw // 8), ("bn.6", 8 * f, 8 * f, 9, hw // 8),
            ("up1", 8 * f, 4 * f, 1, hw // 8), ("c4.0", 8 * f, 4 * f, 9, hw // 4), ("c4.3", 4 * f, 4 * f, 9, hw // 4),
            ("up2", 4 * f, 2 * f, 1, hw // 4), ("c5.0", 4 * f, 2 * f, 9, hw // 2), ("c5.3", 2 * f, 2 * f, 9, hw // 2),
            ("up3", 2 * f, f, 1, hw // 2), ("c6.0", 2 * f, f, 9, hw), ("px", f, nb_classes, 1, hw)]


def step_flops(bs, nb_filters=16, hw=512):
    fwd = sum(2.0 * ci * co * t * h * h for _, ci, co, t, h in unet_conv_table(nb_filters, 3, hw))
    first = 2.0 * 1 * nb_filters * 9 * hw * hw
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
            elif name == "amx_conv2d_dgrad":  # (dpre,Cs@1,wpk,addend,y,Y0s@5,y1,Y1s@7,N@8,H@9,W@10,taps@11,dil,stream)
                fl = 2.0 * args[1] * (args[5] + args[7]) * args[11] * args[8] * args[9] * args[10]
                name = "amx_conv2d_fwd"       # same kernel (conv_fwd_kernel): one family
            elif name == "amx_conv2d_dgrad_fused":   # (dy,aux,k1,k2,k3,bslope,Cs@6,wpk,y,Y0s@9,y1,Y1s@11,N@12,H@13,W@14,taps@15,dil,stream)
                fl = 2.0 * args[6] * (args[9] + args[11]) * args[15] * args[12] * args[13] * args[14]
                name = "amx_conv2d_fwd"       # conv_ws_kernel<.., BWD>: the data gradient with the BatchNorm backward in its loader
            elif name == "amx_conv2d_dgrad_fused_bsum":   # (dy,aux,k1,k2,k3,bslope,Cs@6,wpk,y,Y0s@9,N@10,H@11,W@12,taps@13,dil,bs_a,bs_part,stream)
                fl = 2.0 * args[6] * args[9] * args[13] * args[10] * args[11] * args[12]
                name = "amx_conv2d_fwd"       # conv_ws_kernel<.., BWD, BSUM>: + the BatchNorm-backward sums of the source layer
            elif name == "amx_conv2d_wgrad":  # (.., C0s@3, .., C1s@7, dpre@8, Dos@9, part@10, N@11,H,W,cout@14,taps@15)
                fl = 2.0 * (args[3] + args[7]) * args[14] * args[15] * args[11] * args[12] * args[13]
            elif name == "amx_conv2d_wgrad_fused":  # (.., C0s@3, .., C1s@7, dy@8, .., Dos@14, part, bpart, N@17,H,W,cout@20,taps@21)
                fl = 2.0 * (args[3] + args[7]) * args[20] * args[21] * args[17] * args[18] * args[19]
                name = "amx_conv2d_wgrad"
            d = out.setdefault(name, dict(calls=0, total_ms=0.0, flops=0.0, ms=[]))
            d["calls"] += 1
            d["total_ms"] += ms
            d["flops"] += fl
            d["ms"].append(ms)
        return out

    @staticmethod
    def robust_total_ms(d, ksteps):
        """Sum over the launch positions of a step of the MEDIAN duration over the `ksteps` repetitions.  An event pair
        brackets a launch on the stream, so whenever the host falls behind the GPU (another process on the box, a page
        fault) the idle time until the kernel is enqueued lands inside the pair; a repetition-wise median drops such
        outliers, and equals the mean when there are none (a plain mean once read 0.341 ms per launch while rocprofv3 of
        the same trip — true kernel durations — read 0.317)."""
        ms = d["ms"]
        if ksteps <= 0 or len(ms) % ksteps:
            return d["total_ms"]
        per_step = np.asarray(ms, dtype=np.float64).reshape(ksteps, len(ms) // ksteps)
        return float(np.median(per_step, axis=0).sum() * ksteps)


def physical_cores():
    """Physical core count of the host (unique (package, core) pairs); falls back to os.cpu_count()."""
    try:
        seen = set()
        for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology"):
            seen.add((open(os.path.join(d, "physical_package_id")).read().strip(),
                      open(os.path.join(d, "core_id")).read().strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count()


def cpu_model():
    """`model name` of /proc/cpuinfo ("core count and CPU model printed")."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _one_cpu_per_core(n):
    """The first n logical CPUs that sit on DISTINCT physical cores (no two SMT siblings), for thread confinement."""
    seen, cpus = set(), []
    for d in sorted(glob.glob("/sys/devices/system/cpu/cpu[0-9]*"), key=lambda q: int(q.rsplit("cpu", 1)[1])):
        try:
            key = (open(os.path.join(d, "topology/physical_package_id")).read().strip(),
                   open(os.path.join(d, "topology/core_id")).read().strip())
        except OSError:
            continue
        cpu = int(d.rsplit("cpu", 1)[1])
        if key not in seen and cpu in os.sched_getaffinity(0):
            seen.add(key)
            cpus.append(cpu)
        if len(cpus) == n:
            break
    return cpus


class _Confine:
    """Confines EVERY thread of this process (the OpenMP pool included) to the given CPUs for the timed CPU leg and
    restores the masks afterwards — the 2-step samples of rounds 3-5 wandered 2.7 -> 5.1 -> 6.95 img/s on nominally
    identical hosts; migrating threads and SMT-sibling sharing were part of it."""

    def __init__(self, cpus):
        self.cpus, self.saved = set(cpus), {}

    def __enter__(self):
        if not self.cpus:
            return self
        for t in os.listdir("/proc/self/task"):
            try:
                self.saved[int(t)] = os.sched_getaffinity(int(t))
                os.sched_setaffinity(int(t), self.cpus)
            except OSError:
                pass
        return self

    def __exit__(self, *exc):
        for t, m in self.saved.items():
            try:
                os.sched_setaffinity(t, m)
            except OSError:
                pass
        return False


def cpu_baseline(hw=H, budget_s=60.0):
    """Oracle (CPU restatement, stock PyTorch CPU ops) train step on a bounded sample of the workload, at the BEST
    (batch size, thread count) pair of a small sweep: every thread count is probed at bs 2 AND bs 8 (an oversubscribed
    256-thread run is ~18x slower than 8 threads, and the best thread count at bs 2 is not the best at bs 8), then the
    winner is TIMED: >= 8 steps (budget permitting) with the process's threads confined to one logical CPU per physical
    core, median and spread of the per-step times reported (the 2-step samples of earlier rounds drifted
    2.7 -> 6.95 img/s).  Also the survey's calibration point: U-Net 256^2, bs 8, 8 threads — the
    survey container measured 11.3 img/s on 8 cores there."""
    from oracle import seg_oracle as so
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.rand(8, 1, hw, hw).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, 3, (8, hw, hw)))
    sd = so.init_unet(3, 16, seed=1)
    opt = so.AdamState(lr=1e-3)
    phys = physical_cores()
    cands = sorted({t for t in (8, 16, 32, 64, phys) if t and t <= (os.cpu_count() or 1)})
    t_begin = time.time()
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        for bs in (2, 8):
            so.train_step("Unet", sd, opt, x[:bs], y[:bs], 3)     # warm-up at this (threads, bs)
            t0 = time.time()
            so.train_step("Unet", sd, opt, x[:bs], y[:bs], 3)
            sweep[(bs, t)] = round(bs / (time.time() - t0), 3)
        if time.time() - t_begin > budget_s * 0.4:
            break
    best_bs, best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    xb, yb = x[:best_bs], y[:best_bs]
    per_step = []
    with _Confine(_one_cpu_per_core(best)):
        so.train_step("Unet", sd, opt, xb, yb, 3)                 # warm-up (pool resized, threads confined)
        while len(per_step) < 3 or (len(per_step) < 12 and time.time() - t_begin < budget_s):
            t0 = time.time()
            so.train_step("Unet", sd, opt, xb, yb, 3)
            per_step.append(time.time() - t0)
    med = float(np.median(per_step))
    # calibration against the survey's CPU probe: config-1 shape (256^2, bs 8), 8 threads on 8 distinct cores
    cal = None
    try:
        torch.set_num_threads(8)
        xc = torch.from_numpy(rs.rand(8, 1, 256, 256).astype(np.float32))
        yc = torch.from_numpy(rs.randint(0, 3, (8, 256, 256)))
        with _Confine(_one_cpu_per_core(8)):
            so.train_step("Unet", sd, opt, xc, yc, 3)
            ts = []
            for _ in range(3):
                t0 = time.time()
                so.train_step("Unet", sd, opt, xc, yc, 3)
                ts.append(time.time() - t0)
        cal = {"images_per_s": round(8 / float(np.median(ts)), 2), "threads": 8, "shape": "U-Net 256x256 bs 8 train step",
               "survey_probe_images_per_s": 11.3, "survey_probe_cores": 8}
    except Exception as e:                                   # never let the calibration point take the line down
        cal = {"error": f"{type(e).__name__}: {e}"}
    return {"value": round(best_bs / med, 3), "unit": "images/s", "cores": best, "kind": "port",
            "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "host_physical_cores": phys,
            "steps_timed": len(per_step), "step_s_median": round(med, 4), "step_s_min": round(min(per_step), 4),
            "step_s_max": round(max(per_step), 4),
            "spread_rel": round((max(per_step) - min(per_step)) / med, 4),
            "threads_confined_to": "one logical CPU per physical core",
            "sweep_images_per_s": {f"bs{b}_threads{t}": v for (b, t), v in sweep.items()},
            "calibration_256_bs8": cal,
            "sample": f"{len(per_step)} U-Net train steps (fwd+bwd+Adam) at bs={best_bs}, {hw}x{hw}, fp32, torch CPU ops, "
                      f"{best} threads (the best (bs, threads) pair of the sweep); value = bs / median step time"}


class ClockSampler(threading.Thread):
    """Samples the shader clock (sysfs pp_dpm_sclk, the entry marked '*') once a second while running."""

    def __init__(self, local=0):
        super().__init__(daemon=True)
        self.samples, self._stop_ev = [], threading.Event()
        self.path = None
        try:                                   # the sysfs node of THIS device, found through its PCI address
            pr = torch.cuda.get_device_properties(local)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cand = f"/sys/bus/pci/devices/{bdf}/pp_dpm_sclk"
            if os.path.exists(cand):
                self.path = cand
        except Exception:
            pass
        if self.path is None:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
            self.path = cards[min(local, len(cards) - 1)] if cards else None

    def read(self):
        if not self.path:
            return None
        try:
            for line in open(self.path):
                if "*" in line:
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            return None
        return None

    def run(self):
        while not self._stop_ev.wait(1.0):
            v = self.read()
            if v is not None:
                self.samples.append(v)

    def stop(self):
        self._stop_ev.set()


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one per GPU) from here."""
    port = _free_port()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMX_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # every rank's stderr goes to its own temporary file: the tail of the FIRST rank that fails is what the
        # launcher prints (eight interleaved tracebacks, seven of them "peer closed the connection", tell nothing)
        ef = tempfile.TemporaryFile(mode="w+b")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env,
                                      stderr=ef))
    # Poll instead of waiting rank by rank: a rank that dies leaves the others inside the next collective until the
    # process-group timeout (AMX_DIST_TIMEOUT_S, minutes) — the first non-zero exit ends the job within a second.
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
        elif all(c == 0 for c in codes):
            break
        else:
            time.sleep(0.2)
    if failed is None:
        for r, ef in enumerate(errs):                        # warnings of healthy ranks are still worth seeing
            _relay_stderr(ef, r, n, tail=None)
        return 0
    rc = procs[failed].returncode
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.terminate()
    deadline = time.time() + 5.0
    for p in procs:
        try:
            p.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    print(f"bench.py: rank {failed} of {n} exited with code {rc}; the other ranks were stopped.  "
          f"Its stderr (tail):", file=sys.stderr, flush=True)
    _relay_stderr(errs[failed], failed, n, tail=40)
    return rc if rc > 0 else 1


def _relay_stderr(ef, rank: int, n: int, tail=None) -> None:
    ef.seek(0)
    lines = ef.read().decode(errors="replace").splitlines()
    ef.close()
    if tail is not None:
        lines = lines[-tail:]
    for ln in lines:
        print(f"[rank {rank}/{n}] {ln}", file=sys.stderr)
    sys.stderr.flush()


def extra_configs():
    """Bounded samples of BASELINE.json configs[2..4], each timed by its own harness in tools/bench_extra.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extra as bx
    out = {}
    for key, fn, kw in (("config3_dilnet_predict_4096x1024", bx.bench_predict_full, dict(frames=4096)),
                        ("config4_rvae_bs512_64x64", bx.bench_rvae, dict(steps=20, warmup=3)),
                        ("config5_dkl_rbf_n16384", bx.bench_dkl, dict()),
                        ("config5_dklgpr_fit_step", bx.bench_dkl_fit, dict())):
        t0 = time.perf_counter()
        try:
            r = fn(emit=False, **kw)
            r["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:                               # an extra must never take the headline line down
            r = {"error": f"{type(e).__name__}: {e}"}
        out[key] = r
        torch.cuda.empty_cache()
    return out


class _Mi355x:
    """The device side of this bench: HIP through torch.cuda, RCCL through torch.distributed's "nccl" backend.  (The
    launch-contract tests drive main() with a stand-in of this class from tests/emu/bench_emu.py; bench.py itself has
    no other backend and no switch for one.)"""
    product = True
    dist_backend = None                                      # init_distributed's default on a GPU host: nccl (= RCCL)
    collective = "nccl (RCCL)"

    def check(self, gpus=1):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
        seen = torch.cuda.device_count()
        if seen < gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but only {seen} GPU(s) are visible to this process "
                             f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', 'unset')}, "
                             f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES', 'unset')})")

    def collective_version(self):
        try:
            v = torch.cuda.nccl.version()                    # RCCL reports through torch's nccl binding
            return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as e:                               # never let a version string take the line down
            return f"unknown ({type(e).__name__})"

    def device(self, local):
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def sync(self):
        torch.cuda.synchronize()

    def memory_stats(self, dev):
        return torch.cuda.memory_stats(dev)


def host_enqueue_ms(model, nb, sync, n=3):
    """Host time to ENQUEUE one training step (tools/gpu_host_time.py's measure, inline): zero_grad -> forward -> loss ->
    backward -> [all-reduce] -> Adam with no synchronisation inside; the device is drained before and after.  Every rank
    runs it (the step holds the collective)."""
    ts = []
    for i in range(n):
        feat, tar = model.X_train[i % nb].to(model.device), model.y_train[i % nb].to(model.device)
        sync()
        t0 = time.perf_counter()
        model.net.train()
        model.optimizer.zero_grad()
        kind, out = model.net.forward_loss(feat, tar)      # the fused head + loss node train_step takes (nets/fcnn.py)
        loss = out if kind == "loss" else model.criterion(out, tar)
        loss.backward()
        if model.dp is not None:
            model.dp.allreduce_grads()
        model.optimizer.step()
        ts.append((time.perf_counter() - t0) * 1e3)
        sync()
    return float(np.median(ts))


def xgmi_seen():
    """Does this node report xGMI links between its GPUs?  `rocm-smi --showtopo` (link-type table) when the tool exists:
    {"xgmi_links": n, "pcie_links": m} counted over the GPU pairs, None when the tool is absent or fails."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if not exe:
        return None
    try:
        txt = subprocess.run([exe, "--showtopotype"], capture_output=True, text=True, timeout=30).stdout
        if "XGMI" not in txt and "PCIE" not in txt:
            txt = subprocess.run([exe, "--showtopo"], capture_output=True, text=True, timeout=30).stdout
    except Exception:
        return None
    return {"xgmi_links": txt.count("XGMI") // 2, "pcie_links": txt.count("PCIE") // 2}


def main(argv=None, backend=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="length of the steady-state leg after the timed region (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2..4] one-liners")
    ap.add_argument("--serial", action="store_true",
                    help="run every kernel on one stream (no weight-gradient overlap): the mode of the per-kernel "
                         "HIP-event pass behind `roofline`; used for the rocprofv3 summary that pass must agree with")
    # ---- test-only overrides (a line produced with any of them carries "headline": false)
    ap.add_argument("--hw", type=int, default=H, help="TEST ONLY: image size")
    ap.add_argument("--bs", type=int, default=BS, help="TEST ONLY: batch size per GPU")
    ap.add_argument("--nb-filters", type=int, default=16, help="TEST ONLY: U-Net width")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="TEST ONLY: this rank raises after the warm-up (the launcher must end the job at once)")
    args = ap.parse_args(argv)
    be = backend or _Mi355x()

    if int(os.environ.get("WORLD_SIZE", str(args.gpus))) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
    # (before any rank is spawned: one clear line, not N tracebacks.  Under a launcher a rank needs the devices of ITS node
    #  only: LOCAL_WORLD_SIZE — a 2-node x 8-GPU torchrun job passes --gpus 16 with 8 visible devices per process)
    be.check(int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus)) if "WORLD_SIZE" in os.environ else args.gpus)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    import atomai_amd as aoi
    from atomai_amd.parallel import DataParallelGrads, init_distributed
    force_dp = os.environ.get("AMX_BENCH_FORCE_DP") == "1"      # N=1 through the RCCL branch (a 1-rank nccl group)
    rank, world, local = init_distributed(be.dist_backend, force=force_dp)
    hw, bs = args.hw, args.bs
    headline = (hw, bs, args.nb_filters) == (H, BS, 16) and be.product
    dev = be.device(local)
    sync = be.sync

    rs = np.random.RandomState(rank)                         # a different shard per rank
    nb = 2                                                   # distinct mini-batches resident per GPU
    X = rs.rand(nb * bs, hw, hw).astype(np.float32)
    y = rs.randint(0, 3, (nb * bs, hw, hw))
    model = aoi.models.Segmentor("Unet", nb_classes=3, seed=1, nb_filters=args.nb_filters)
    model.compile_trainer((X, y, X[:bs], y[:bs]), loss="ce", training_cycles=args.steps + args.warmup,
                          batch_size=bs, plot_training_history=False)
    if world > 1 or force_dp:
        model.dp = DataParallelGrads(model.optimizer, model.net)
    if world > 1 or force_dp:
        model.dp.timing = True                               # HIP events around the gradient all-reduce of every step
    timer = None if (args.no_kernel_timing or not be.product) else KernelTimer(
        ["amx_conv2d_fwd", "amx_conv2d_dgrad", "amx_conv2d_dgrad_fused", "amx_conv2d_dgrad_fused_bsum", "amx_conv2d_wgrad",
         "amx_conv2d_wgrad_fused"])
    from atomai_amd.engine import Tape
    if args.serial:
        Tape.use_side_stream = False

    def barrier():
        if world > 1 or force_dp:
            torch.distributed.barrier()
        sync()

    def one_step(i):
        return model.train_step(model.X_train[i % nb], model.y_train[i % nb])[0]

    losses = []
    wmarks = [time.perf_counter()]
    for i in range(args.warmup):
        losses.append(one_step(i))
        wmarks.append(time.perf_counter())
    if args.fail_rank == rank:
        raise RuntimeError(f"--fail-rank {rank}: injected failure (launcher test)")
    barrier()
    ms0 = be.memory_stats(dev)
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(args.steps):
        losses.append(one_step(i))
        marks.append(time.perf_counter())                    # host time at which the step's loss value arrived
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = np.diff(marks) * 1e3
    ms1 = be.memory_stats(dev)
    alloc_info = {"hipMalloc_calls_in_timed_region": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                  "hipFree_calls_in_timed_region": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                  "reserved_GB": round(ms1.get("reserved_bytes.all.current", 0) / 1e9, 2),
                  "peak_allocated_GB": round(ms1.get("allocated_bytes.all.peak", 0) / 1e9, 2)}
    elapsed_own = elapsed
    if world > 1 or force_dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sustained leg: the same step for >= sustain-seconds (every rank runs the same, pre-agreed step count)
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds / (elapsed / args.steps))))
        clk = ClockSampler(local)
        clk0 = clk.read()
        clk.start()
        barrier()
        s0 = time.perf_counter()
        smarks = [s0]
        for i in range(n_sus):
            one_step(i)
            smarks.append(time.perf_counter())
        barrier()
        s_el = time.perf_counter() - s0
        clk.stop()
        if world > 1 or force_dp:
            t = torch.tensor([s_el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            s_el = float(t.item())
        sp = np.diff(smarks) * 1e3
        sustained = {"seconds": round(s_el, 2), "steps": n_sus, "images_per_s": round(world * bs * n_sus / s_el, 2),
                     "ms_per_step": round(s_el / n_sus * 1e3, 3),
                     "step_ms_min_median_max": [round(float(sp.min()), 3), round(float(np.median(sp)), 3),
                                                round(float(sp.max()), 3)],
                     "step_ms_p99": round(float(np.percentile(sp, 99)), 3),
                     "sclk_mhz_idle_then_samples": ([clk0] + clk.samples) if clk0 is not None else None,
                     "_sclk_own": [float(v) for v in clk.samples]}

    # ---- per-rank diagnosis of a multi-GPU run (what a first real 8-GPU line is read by): every rank's
    # own wall time per step, the time of its gradient all-reduces (events around the collective; percentiles over all
    # ranks and steps), the host time it needs to ENQUEUE one step (eight launch threads share one host), its shader clock
    # under load, and whether the node reports xGMI links at all.  How to read it: DESIGN.md section 4 "First 8-GPU run".
    dp_info = None
    if world > 1 or force_dp:
        ar = model.dp.allreduce_ms()[args.warmup:args.warmup + args.steps]        # the timed region's collectives
        host_ms = host_enqueue_ms(model, nb, sync)
        clk_s = sorted(sustained["_sclk_own"]) if (sustained and sustained.get("_sclk_own")) else []
        clk3 = [clk_s[0], clk_s[len(clk_s) // 2], clk_s[-1]] if clk_s else [0.0, 0.0, 0.0]
        arv = list(ar) + [0.0] * (args.steps - len(ar))
        mine = torch.tensor([elapsed_own / args.steps * 1e3, float(np.mean(ar)) if ar else 0.0,
                             float(np.max(ar)) if ar else 0.0, host_ms] + clk3 + arv, device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        ar_all = allr[:, 7:].reshape(-1)
        steps_ms = allr[:, 0]
        dp_info = {"per_rank_ms_per_step": [round(float(v), 3) for v in steps_ms],
                   "per_rank_spread_rel": round(float((steps_ms.max() - steps_ms.min()) / steps_ms.mean()), 4),
                   "allreduce_ms": round(float(allr[:, 1].mean()), 4),
                   "allreduce_ms_per_rank_mean": [round(float(v), 4) for v in allr[:, 1]],
                   "allreduce_ms_per_rank_max": [round(float(v), 4) for v in allr[:, 2]],
                   "allreduce_ms_percentiles": {k: round(float(np.percentile(ar_all, q)), 4)
                                                for k, q in (("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
                   "allreduce_bytes": int(model.optimizer._flat["g"].numel() * 4),
                   "allreduce_note": "one sum all-reduce of the flat fp32 gradient bucket per step; events on the "
                                     "launch stream right before / after dist.all_reduce (the wait for backward is "
                                     "not inside the pair)",
                   "host_enqueue_ms_per_step_per_rank": [round(float(v), 3) for v in allr[:, 3]],
                   "host_enqueue_note": "host time to enqueue one whole step (zero_grad .. optimizer.step, no "
                                        "synchronisation inside), median of 3 probe steps after the timed regions",
                   "sclk_mhz_per_rank_min_median_max": ([[int(v) for v in row[4:7]] for row in allr]
                                                        if float(allr[:, 4:7].max()) > 0 else None),
                   "xgmi_seen": xgmi_seen() if rank == 0 else None}
    if sustained:
        sustained.pop("_sclk_own", None)

    ksteps = 0
    if timer:                      # every rank takes part (the step contains the gradient all-reduce)
        # Per-kernel HIP-event pass, in the same run after the timed regions: the step normally overlaps
        # the weight-gradient kernels with HBM-bound kernels on a second stream, which makes per-launch event
        # durations meaningless, so this pass serialises everything on one stream (it does not enter `value`).
        Tape.use_side_stream = False
        ksteps = min(args.steps, 5)
        one_step(0)                # untimed: the serial schedule allocates differently (first touch of fresh blocks)
        sync()
        timer.active = True
        for i in range(ksteps):
            one_step(i)
        sync()
        timer.active = False
        Tape.use_side_stream = not args.serial
    barrier()
    if rank != 0:
        if world > 1:
            torch.distributed.barrier()              # rank 0 may still be formatting; leave together
        return
    ms = elapsed / args.steps * 1e3
    value = world * bs * args.steps / elapsed
    fl = step_flops(bs, args.nb_filters, hw)
    out = {
        "metric": "training images/sec (512x512, bs=32/GPU) U-Net Segmentor",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Segmentor U-Net nb_classes=3, {hw}x{hw}, bs={bs}/GPU, fp32, CE loss, Adam 1e-3 "
                               "(BASELINE.json configs[1]); step = fwd+bwd+allreduce+Adam+loss.item()",
                   "global_batch": world * bs, "parallelism": f"dp{world}",
                   "world_size_seen": (torch.distributed.get_world_size() if torch.distributed.is_initialized()
                                       else world),          # what the process group itself reports
                   "collective_backend": be.collective if (world > 1 or force_dp) else None,
                   "rccl_version": be.collective_version() if (world > 1 or force_dp) else None,
                   "cpu_threads_per_rank": torch.get_num_threads(),
                   "launcher": "self" if os.environ.get("AMX_BENCH_SELF_LAUNCHED") else
                               ("torchrun" if "WORLD_SIZE" in os.environ else "single"),
                   "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)]},
        "headline": headline,
        "step_ms_median_max": [round(float(np.median(per_step)), 3), round(float(per_step.max()), 3)],
        "allocator": alloc_info,
        "step_ms_all": [round(float(v), 1) for v in list(np.diff(wmarks) * 1e3) + list(per_step)],
        "step_ms_note": "host-side intervals between loss values: the loss of step t reaches the host after its forward "
                        "pass (trainer._EarlyScalar), its backward + Adam overlap the host work of step t+1; the timed "
                        "region is closed by a device synchronisation, ms_per_step = region / steps",
        "step_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
        "step_frac_of_mfma_f32_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_MFMA_F32, 4),
    }
    if not be.product:
        out["backend"] = be.collective + " — tests only; not a measurement"
    if dp_info:
        out.update(dp_info)
    if sustained:
        sustained["agrees_with_value_within"] = round(abs(sustained["images_per_s"] / value - 1.0), 4)
        out["sustained"] = sustained
    if timer:
        summ = timer.summarize()
        conv = summ.get("amx_conv2d_fwd")
        if conv:
            conv_mean_ms = conv["total_ms"]
            conv["total_ms"] = KernelTimer.robust_total_ms(conv, ksteps)
            ach = conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12
            traffic, tsrc = None, None       # HBM bytes per launch: rocprofv3 PMC pass committed under profiles/
            pmc = os.path.join(ROOT, PMC_FILE)
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                traffic = pj["conv_fwd_family"]["hbm_MB_per_launch"] * 1e6
                tsrc = f"{PMC_FILE}@{pj.get('git_head', 'unknown')} (separate rocprofv3 --pmc pass, not this run)"
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32, 4), "traffic": traffic,
                               "traffic_source": tsrc,
                               "kernel": "conv_fwd_kernel<TAPS,NT,HALO> + conv_ws_kernel<NCH,NT,BWD,BSUM> (amx_conv2d_fwd / amx_conv2d_dgrad / "
                                         "amx_conv2d_dgrad_fused[_bsum]: all forward + dgrad launches of the step)",
                               "launches_per_step": conv["calls"] // ksteps,
                               "ms_per_step": round(conv["total_ms"] / ksteps, 3),
                               "avg_launch_ms": round(conv["total_ms"] / conv["calls"], 4),
                               "avg_launch_ms_plain_mean": round(conv_mean_ms / conv["calls"], 4),
                               "measured": f"HIP events on the launch stream, {ksteps} serialised steps after the timed region; "
                                           "per launch position the median over the steps (drops host-induced gaps inside "
                                           "an event pair)"}
        wg = summ.get("amx_conv2d_wgrad")
        if wg:
            wg["total_ms"] = KernelTimer.robust_total_ms(wg, ksteps)
            ach = wg["flops"] / (wg["total_ms"] * 1e-3) / 1e12
            out["roofline_wgrad"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32,
                                     "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32, 4),
                                     "kernel": "wgrad_ws_kernel<NT,WM,WN,TH> (wave-specialised, <= 32 input channels) + wgrad_kernel<TAPS,NT,WM,HALO> "
                                               "(amx_conv2d_wgrad_fused: all weight-gradient launches of the step)",
                                     "ms_per_step": round(wg["total_ms"] / ksteps, 3)}
    if world == 1 and be.product:
        del model
        torch.cuda.empty_cache()
        if not args.no_extra:
            out["extra_configs"] = extra_configs()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
