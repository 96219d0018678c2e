"""Instrumented copy of SegPredictor.batch_predict's chunk loop: where does the host spend its time per chunk?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd.engine import aux_stream
torch.manual_seed(1)
net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
rs = np.random.RandomState(0)
frames = 256
stack = rs.rand(frames, 1024, 1024).astype(np.float32)
p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
p.run(stack[:16], compute_coords=False)
for chunk_mb in (64, 256):
    p.chunk_bytes = chunk_mb << 20
    data = p.preprocess(stack, True)
    n = len(data); out_shape = (n, 1024, 1024, 1)
    T = dict(alloc=0, drain_wait=0, drain_copy=0, pin_in=0, launch=0)
    torch.cuda.synchronize(); t_all = time.perf_counter()
    t = time.perf_counter(); out = torch.empty(out_shape); T["alloc"] += time.perf_counter() - t
    chunk = max(1, min(n, p.chunk_bytes // (4 << 20)))
    dev = torch.device("cuda")
    copy_in, copy_out = aux_stream(dev, 1), aux_stream(dev, 2)
    main = torch.cuda.current_stream(dev)
    pin_in = [torch.empty((chunk, 1, 1024, 1024), pin_memory=True) for _ in range(2)]
    pin_out = [torch.empty((chunk, 1024, 1024, 1), pin_memory=True) for _ in range(2)]
    pending = [None, None]
    gpu_ev = []
    mallocs = []
    def drain(slot):
        if pending[slot] is not None:
            ev, ps, pm = pending[slot]
            t = time.perf_counter(); ev.synchronize(); T["drain_wait"] += time.perf_counter() - t
            t = time.perf_counter(); out[ps:ps + pm] = pin_out[slot][:pm]; T["drain_copy"] += time.perf_counter() - t
            pending[slot] = None
    for k, s in enumerate(range(0, n, chunk)):
        slot, m = k % 2, min(chunk, n - s)
        drain(slot)
        t = time.perf_counter(); pin_in[slot][:m].copy_(data[s:s + m]); T["pin_in"] += time.perf_counter() - t
        t = time.perf_counter()
        with torch.cuda.stream(copy_in):
            d = pin_in[slot][:m].to(dev, non_blocking=True)
            ev_in = torch.cuda.Event(); ev_in.record(copy_in)
        main.wait_event(ev_in)
        ta = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main)
        prob = p.forward_(d)
        e1.record(main); gpu_ev.append((e0, e1))
        tb = time.perf_counter()
        done = torch.cuda.Event(); done.record(main)
        with torch.cuda.stream(copy_out):
            copy_out.wait_event(done)
            pin_out[slot][:m].copy_(prob, non_blocking=True)
            ev_out = torch.cuda.Event(); ev_out.record(copy_out)
        pending[slot] = (ev_out, s, m)
        T["launch"] += time.perf_counter() - t
        st = torch.cuda.memory_stats()
        mallocs.append((st["num_device_alloc"], st["num_device_free"], round(st["reserved_bytes.all.current"] / 1e9, 1),
                        round((ta - t) * 1e3, 1), round((tb - ta) * 1e3, 1), round((time.perf_counter() - tb) * 1e3, 1)))
    drain(0); drain(1)
    torch.cuda.synchronize(); total = time.perf_counter() - t_all
    gpu = sum(a.elapsed_time(b) for a, b in gpu_ev)
    gaps = [gpu_ev[i][1].elapsed_time(gpu_ev[i + 1][0]) for i in range(len(gpu_ev) - 1)]
    print("  per chunk (hipMalloc, hipFree, reserved GB, host ms: H2D issue, forward launches, D2H issue):", mallocs[:10])
    print("  main-stream gaps between consecutive chunks (ms):", [round(g, 2) for g in gaps][:8])
    print(f"chunk {chunk_mb} MB ({chunk} frames): total {total/n*1e3:.3f} ms/frame | GPU forward {gpu/n:.3f} ms/frame | host per frame: "
          + ", ".join(f"{k} {v/n*1e3:.3f}" for k, v in T.items()), flush=True)
