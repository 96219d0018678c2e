"""Round 4 probe behind the full-size parity tests: (b) per-parameter gradient error of the default-width nets against the
fp64 oracle next to the oracle's own fp32 floor and the LeakyReLU kink sensitivity; (a) 3 Adam steps of the bs-32 512^2
U-Net against oracle.seg_oracle.train_step on the device."""
import os, sys, json
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from oracle import seg_oracle as so


def on(device, sd, x, y, ncls, model, **kw):
    sd = OrderedDict((k, v.to(device)) for k, v in sd.items())
    return so.loss_and_grads(model, sd, x.to(device), y.to(device), ncls, **kw)


def grads_probe(model, ncls, B, H, tau=1e-5):
    torch.manual_seed(1)
    net, _ = aoi.nets.init_fcnn_model(model, ncls)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.rand(B, 1, H, H).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, ncls, (B, H, H))) if ncls > 1 else torch.from_numpy((rs.rand(B, 1, H, H) > 0.5).astype(np.float32))
    net.cuda().train()
    crit = aoi.losses_metrics.select_loss("ce", ncls)
    loss = crit(net(x.cuda()), y.cuda()); loss.backward()
    _, _, g32 = on("cuda", sd, x, y, ncls, model)
    y64 = y if ncls > 1 else y.double()
    sd64 = so.cast(sd, torch.float64)
    _, _, g64 = on("cuda", sd64, x.double(), y64, ncls, model)
    so.KINK_FLIP = [tau, 0]
    try:
        _, _, gfl = on("cuda", sd64, x.double(), y64, ncls, model)
        nflip = so.KINK_FLIP[1]
    finally:
        so.KINK_FLIP = None
    gmax = max(float(g.abs().max()) for g in g64.values())
    rows = []
    for k, p in net.named_parameters():
        err = float((p.grad.double() - g64[k]).abs().max()) / gmax
        floor = float((g32[k].double() - g64[k]).abs().max()) / gmax
        sens = float((gfl[k] - g64[k]).abs().max()) / gmax
        rows.append((k, err, floor, sens))
    worst = max(rows, key=lambda r: r[1])
    print(f"{model} B={B} H={H}: {nflip} LeakyReLU inputs within {tau:g} of 0; worst err {worst[1]:.2e} ({worst[0]}, floor {worst[2]:.2e}, "
          f"kink sens {worst[3]:.2e}); max floor {max(r[2] for r in rows):.2e}; max sens {max(r[3] for r in rows):.2e}; "
          f"max err/(floor) {max(r[1] / max(r[2], 1e-12) for r in rows):.2f}; #err>1e-4: {sum(r[1] > 1e-4 for r in rows)} of {len(rows)}", flush=True)
    for r in sorted(rows, key=lambda r: -r[1])[:5]:
        print(f"     {r[0]:40s} err {r[1]:.2e} floor {r[2]:.2e} sens {r[3]:.2e}")


def trajectory():
    rs = np.random.RandomState(0)
    X = rs.rand(32, 512, 512).astype(np.float32); y = rs.randint(0, 3, (32, 512, 512))
    m = aoi.models.Segmentor(nb_classes=3, seed=1)
    m.compile_trainer((X, y, X, y), training_cycles=3, batch_size=32)
    sd0 = OrderedDict((k, v.detach().clone()) for k, v in m.net.state_dict().items())
    ls = [m.train_step(m.X_train[0], m.y_train[0])[0] for _ in range(3)]
    xb, yb = m.X_train[0].detach().clone(), m.y_train[0].detach().clone()
    out = {"hip": ls}
    for name, dt in (("oracle_f32", torch.float32), ("oracle_f64", torch.float64)):
        sd = OrderedDict((k, (v.to(dt) if v.dtype.is_floating_point else v.clone()).cuda()) for k, v in sd0.items())
        opt = so.AdamState(lr=1e-3)
        out[name] = [so.train_step("Unet", sd, opt, xb.to(dt).cuda(), yb.cuda(), 3) for _ in range(3)]
        torch.cuda.empty_cache()
    print(json.dumps(out))
    for i in range(3):
        print(f"step {i + 1}: hip {out['hip'][i]:.7f} f32 {out['oracle_f32'][i]:.7f} f64 {out['oracle_f64'][i]:.7f}  "
              f"rel(hip,f64) {abs(out['hip'][i] - out['oracle_f64'][i]) / out['oracle_f64'][i]:.2e}  rel(f32,f64) {abs(out['oracle_f32'][i] - out['oracle_f64'][i]) / out['oracle_f64'][i]:.2e}")
    print("peak GB", torch.cuda.max_memory_allocated() / 1e9)


for a in [("Unet", 3, 4, 512), ("dilnet", 1, 2, 256), ("SegResNet", 3, 4, 256), ("ResHedNet", 3, 2, 128)]:
    grads_probe(*a)
    torch.cuda.empty_cache()
trajectory()
