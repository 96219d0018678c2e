"""When does the host learn that a side-stream kernel finished while the main stream keeps computing?  (dev tool)
Main stream: a long chain of convolution work A, then B.  Side stream: waits for A, runs a copy kernel into pinned
host memory, records an event.  The host measures when event.synchronize() returns relative to A's and B's ends."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd import _lib as L
from atomai_amd.nets.fcnn import predict_proba

torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
net = net.cuda().eval()
x = torch.rand(16, 1, 1024, 1024, device="cuda")
side = torch.cuda.Stream()
pin = torch.empty(16, 1024, 1024, 1, pin_memory=True)
for _ in range(2): predict_proba(net, x)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
for mode in ("kernel", "memcpy"):
    for timing in (False, True):
        res = []
        for rep in range(3):
            a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True); b1 = torch.cuda.Event(enable_timing=True)
            a0.record()
            pa = predict_proba(net, x).contiguous()           # A: ~23 ms
            a1.record()
            side.wait_event(a1)
            if mode == "kernel":
                L.call("amx_copy16", L.ptr(pa), ctypes.c_void_p(pin.data_ptr()), pa.numel() * 4, 128, ctypes.c_void_p(side.cuda_stream))
            else:
                with torch.cuda.stream(side):
                    pin.copy_(pa, non_blocking=True)
            ev = torch.cuda.Event(enable_timing=timing)
            ev.record(side)
            pb = predict_proba(net, x); pb2 = predict_proba(net, x)   # B: ~46 ms more on the main stream
            b1.record()
            t0 = time.perf_counter()
            ev.synchronize()
            t_ev = time.perf_counter() - t0
            b1.synchronize()
            t_b = time.perf_counter() - t0
            res.append((a0.elapsed_time(a1), a0.elapsed_time(b1), 1e3 * t_ev, 1e3 * t_b))
        print(f"{mode:6s} timing={timing}: " + " | ".join(f"A {a:.1f} ms, A+B {ab:.1f} ms; host: side event after {te:.1f} ms, B after {tb:.1f} ms" for a, ab, te, tb in res), flush=True)
