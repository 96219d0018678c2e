"""CU partitioning between the weight-gradient side stream and the main dependency chain of the U-Net backward
(VERDICT r05 item 1): interleaved in-process A/B of the bs-32 512^2 training step with the side stream created by
hipExtStreamCreateWithCUMask (k CUs, two bit layouts), the persistent weight-gradient grids sized to the mask
(AMX_WGRAD_WGS = k), and optionally the whole main chain on the complementary mask.

  python tools/gpu_cumask_ab.py [variant ...]     variant = side:<layout><k>[,wgs:<n>][,main:t|comp|<layout><k>]
     layout `f` = the first k mask bits (k / 8 CUs of every XCD — the KFD deals mask bits round-robin over the XCDs,
     tools/micro/cumask_probe.hip: honoured), layout `x` = k / 32 whole XCDs (the probe shows such a mask is IGNORED:
     all 256 CUs);  `base` = the product (torch's default stream + an unmasked non-blocking side stream).
     main:t = the whole step on a non-blocking torch stream.  hipExtStreamCreateWithCUMask makes a BLOCKING stream
     (it synchronises with the legacy null stream, which is torch's default stream), so every masked variant must keep the
     main chain off the null stream — the first run of this tool did not and measured 28.7 ms for a FULL mask.
"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd import _lib as L, engine

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def mask_bits(spec):
    layout, k = spec[0], int(spec[1:])
    bits = [(i < k) if layout == "f" else ((i % 8) < k // 32) for i in range(256)]
    assert sum(bits) == k, spec
    return bits


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


DEFAULT = ["base", "main:t", "main:t,side:f256,wgs:256", "main:t,side:f224,wgs:224", "main:t,side:f192,wgs:192",
           "main:t,side:f128,wgs:128", "main:t,side:f96,wgs:96", "main:t,side:f64,wgs:64", "main:t,side:f128,wgs:256",
           "main:t,side:f192,wgs:256", "main:comp,side:f128,wgs:128", "main:comp,side:f96,wgs:96", "main:comp,side:f64,wgs:64",
           "main:f224,side:f256,wgs:256"]
variants = sys.argv[1:] or DEFAULT
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
dev = torch.device("cuda", torch.cuda.current_device())
plain_side = engine.aux_stream(dev, 0)
streams = {}


def configure(v):
    """-> the stream the whole step is enqueued on (None = torch's current stream)"""
    kv = dict(x.split(":") for x in v.split(",") if ":" in x)
    L.set_knob("AMX_WGRAD_WGS", kv.get("wgs"))
    side = plain_side
    if "side" in kv:
        if ("s", kv["side"]) not in streams:
            streams[("s", kv["side"])] = masked_stream(mask_bits(kv["side"]))
        side = streams[("s", kv["side"])]
    engine._SIDE_STREAMS[(dev.index, 0)] = side
    main = None
    if "main" in kv:
        key = ("m", kv["main"], kv.get("side") if kv["main"] == "comp" else None)
        if key not in streams:
            if kv["main"] == "t":
                streams[key] = torch.cuda.Stream()
            else:
                bits = [not b for b in mask_bits(kv["side"])] if kv["main"] == "comp" else mask_bits(kv["main"])
                streams[key] = masked_stream(bits)
        main = streams[key]
    return main


def steps(n, main):
    if main is None:
        for i in range(n): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
    else:
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            for i in range(n): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        torch.cuda.current_stream().wait_stream(main)


res = {v: [] for v in variants}
for rep in range(3):
    for v in variants:
        main = configure(v)
        steps(3, main)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps(8, main)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 8 * 1e3)
configure("base")
for k, v in res.items():
    print(f"{k:40s} step ms {['%.2f' % t for t in v]}  min {min(v):.3f}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/cumask_ab.json", "w"), indent=1)
