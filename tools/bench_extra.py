"""Secondary workloads of BASELINE.json (configs[2..4]) — one JSON line each, written to gpurun_out/.
   python tools/bench_extra.py rvae|predict|dkl
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atomai_amd as aoi
from atomai_amd import _lib as L

PEAK = 157.3


def timed_calls(names):
    recs, orig = [], L.call

    def call(name, *a):
        if name not in names:
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(name, *a); e1.record()
        recs.append((name, e0, e1))
        return r
    L.call = call
    import atomai_amd.nets.ed as ed, atomai_amd.engine as eng
    ed.L.call = call; eng.L.call = call
    return recs


def bench_rvae(steps=10, warmup=3, B=512, emit=True, ab=True):
    """configs[3] on one GPU.  Rooflines are priced on the ALGORITHMIC FLOPs of SURVEY.md section 8-d — forward
    2*NL*HID^2 per pixel (139 GFLOP at bs 512), backward = dgrad + wgrad = 2x that (278), step = 3x (417) — whatever the
    backward kernel ISSUES (the recompute variant issues 3x forward; that extra work is not "achieved").  `ab`: the
    same step with the hidden activations recomputed in backward instead of saved by the forward (in-process A/B)."""
    import atomai_amd.nets.ed as ed
    rs = np.random.RandomState(0)
    X = rs.rand(B * 2, 64, 64).astype(np.float32)
    m = aoi.models.rVAE((64, 64), latent_dim=2, seed=0)
    m.dx_prior, m.kdict_["phi_prior"] = 0.1, 0.1
    m.compile_trainer((X, None), None, batch_size=B)
    xs = [torch.from_numpy(X[i * B:(i + 1) * B]).cuda() for i in range(2)]
    names = {"amx_rdecoder_fwd", "amx_rdecoder_bwd", "amx_rdecoder_fwd_save", "amx_rdecoder_bwd_saved"}
    recs = timed_calls(names)

    from atomai_amd.trainers.trainer import _EarlyScalar

    def step(i):                                      # the body of viBaseTrainer.train_epoch for one mini-batch
        m.optim.zero_grad()
        elbo = m.forward_compute_elbo(xs[i % 2])
        early = _EarlyScalar(elbo)
        (-elbo).backward()
        m.optim.step()
        return early.item()

    def run():
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize(); recs.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            last = step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tf = sum(e0.elapsed_time(e1) for n, e0, e1 in recs if "fwd" in n) / steps
        tb = sum(e0.elapsed_time(e1) for n, e0, e1 in recs if "bwd" in n) / steps
        return dt, tf, tb, last
    saved_default = ed.RDEC_SAVE[0]
    dt, tf, tb, last = run()
    rows = B * 64 * 64
    fl_fwd = rows * 2 * 2 * 128 * 128.0
    tfl = lambda fl, ms: round(fl / (ms * 1e-3) / 1e12, 2)          # noqa: E731
    out = {"metric": "rVAE training patches/sec (64x64, bs=512, latent_dim=2)", "value": round(B / dt, 1),
           "unit": "patches/s", "ms_per_step": round(dt * 1e3, 3), "n_gpus": 1, "dtype": "f32", "elbo": last,
           "backward_variant": "saved activations" if saved_default else "recompute",
           "flops_note": "algorithmic (SURVEY 8-d): fwd 139 GFLOP, bwd 278, step 417 at bs 512",
           "roofline": {"bound": "mfma", "kernel": "rdecoder_bwd_kernel<128,64,2>",
                        "achieved": tfl(2 * fl_fwd, tb), "peak": PEAK, "unit": "TFLOP/s",
                        "frac": round(tfl(2 * fl_fwd, tb) / PEAK, 4), "ms": round(tb, 3)},
           "roofline_fwd": {"kernel": "rdecoder_fwd_kernel<128,64>", "achieved": tfl(fl_fwd, tf),
                            "frac": round(tfl(fl_fwd, tf) / PEAK, 4), "ms": round(tf, 3)},
           "roofline_decoder_pair": {"achieved": tfl(3 * fl_fwd, tf + tb), "frac": round(tfl(3 * fl_fwd, tf + tb) / PEAK, 4),
                                     "ms": round(tf + tb, 3)},
           "step_frac_of_mfma_f32_peak": round(tfl(3 * fl_fwd, dt * 1e3) / PEAK, 4)}
    if ab and not os.environ.get("AMX_RVAE_NO_AB"):
        ed.RDEC_SAVE[0] = not saved_default
        try:
            dt2, tf2, tb2, _ = run()
        finally:
            ed.RDEC_SAVE[0] = saved_default
        out["ab_other_variant"] = {"backward_variant": "recompute" if saved_default else "saved activations",
                                   "ms_per_step": round(dt2 * 1e3, 3), "fwd_ms": round(tf2, 3), "bwd_ms": round(tb2, 3),
                                   "patches_per_s": round(B / dt2, 1)}
    if emit:
        print(json.dumps(out), flush=True)
    return out


def bench_predict(frames=256, hw=1024, emit=True):
    """configs[2] on a bounded stack: dilnet nb_classes=1 predict over `frames` 1024x1024 frames."""
    torch.manual_seed(1)
    net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
    rs = np.random.RandomState(0)
    stack = rs.rand(frames, hw, hw).astype(np.float32)
    p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
    p.run(stack[:min(frames, 16)], compute_coords=False)     # warm-up with a full chunk: the pinned staging buffers get their final size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = p.run(stack, compute_coords=False)
    dt = time.perf_counter() - t0
    x = torch.from_numpy(stack[:8, None]).cuda()
    from atomai_amd.nets.fcnn import predict_proba
    net.eval()
    for _ in range(2): predict_proba(net, x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5): predict_proba(net, x)
    torch.cuda.synchronize(); dk = (time.perf_counter() - t1) / 5 / 8
    res = {"metric": "dilnet predict frames/sec (1024x1024, nb_classes=1), end-to-end incl. H2D/D2H",
           "value": round(frames / dt, 2), "unit": "frames/s", "frames": frames,
           "device_only_ms_per_frame": round(dk * 1e3, 3),
           "device_tflops": round(91.62e9 / dk / 1e12, 2), "device_frac_of_mfma_peak": round(91.62e9 / dk / 1e12 / PEAK, 4),
           "out_shape": list(out.shape)}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_predict_full(frames=4096, hw=1024, emit=True):
    """BASELINE.json configs[2] at its FULL size: dilnet nb_classes=1, model.predict over a 4096-frame 1024x1024 stack
    (17.2 GB in, 17.2 GB out, through pinned staging buffers; compute_coords=False).  The stack is filled block by
    block from 64 random frames with a per-block gain (so global min / max live in different blocks); reported:
    end-to-end seconds / frames/s of the whole call and the steady rate (4096-frame call minus a 256-frame call)."""
    torch.manual_seed(1)
    net, _ = aoi.nets.init_fcnn_model("dilnet", 1)
    rs = np.random.RandomState(0)
    base = torch.from_numpy(rs.rand(64, hw, hw).astype(np.float32))
    t0 = time.perf_counter()
    stack = np.empty((frames, hw, hw), dtype=np.float32)
    st = torch.from_numpy(stack)
    for i in range(0, frames, 64):
        m = min(64, frames - i)
        torch.mul(base[:m], 0.5 + (i // 64) / 128.0, out=st[i:i + m])
    t_gen = time.perf_counter() - t0
    p = aoi.predictors.SegPredictor(net, use_gpu=True, nb_classes=1, downsampling=2, verbose=False)
    p.run(stack[:16], compute_coords=False)                   # warm-up with a full chunk (pinned staging buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.run(stack[:256], compute_coords=False)
    t256 = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = p.run(stack, compute_coords=False)
    dt = time.perf_counter() - t0
    chk = float(out[::257].mean())
    # Parity of the timed call's OUTPUT (outside the timed region): 8 frames spread over the stack — first, last, and both
    # sides of chunk borders (16-frame chunks, 3 in flight, collected by a worker thread) — against the oracle's eval
    # graph (oracle.seg_oracle.predict_probs: stock torch ops, fp64, on the device) on the same globally normalised input
    # (utils/preproc.py:torch_format_image: (x - min) / ptp over the WHOLE stack, numpy float32 arithmetic).
    from oracle import seg_oracle as so
    idx = sorted({i for i in (0, 15, 16, frames // 2 - 1, frames // 2, frames - 17, frames - 16, frames - 1) if 0 <= i < frames})
    mn, mx = (float(v) for v in torch.aminmax(st))
    sub = (stack[idx] - np.float32(mn)) / np.float32(mx - mn)
    sd64 = so.cast({k: v.detach().cpu() for k, v in net.state_dict().items()}, torch.float64)
    sd64 = {k: v.cuda() for k, v in sd64.items()}
    ref = so.predict_probs("dilnet", sd64, torch.from_numpy(sub[:, None]).double().cuda(), 1).cpu().numpy()
    got = out[idx].astype(np.float64)
    parity = float(np.abs(got - ref).max() / np.abs(ref).max())
    del out, sd64
    x = torch.from_numpy(stack[:16, None]).cuda()            # device-only rate of the network itself (16-frame chunk)
    from atomai_amd.nets.fcnn import predict_proba
    net.eval()
    for _ in range(2): predict_proba(net, x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5): predict_proba(net, x)
    torch.cuda.synchronize(); dk = (time.perf_counter() - t1) / 5 / 16
    res = {"metric": "dilnet predict over the full 4096-frame 1024x1024 stack (BASELINE configs[2]), end-to-end incl. "
                     "host min/max, H2D, D2H and the copy into the returned array",
           "value": round(frames / dt, 2), "unit": "frames/s", "frames": frames, "seconds": round(dt, 3),
           "steady_frames_per_s": round((frames - 256) / (dt - t256), 2) if frames > 256 else None,
           "first_256_frames_s": round(t256, 3), "stack_GB": round(stack.nbytes / 1e9, 2),
           "host_fill_s": round(t_gen, 2), "out_shape": [frames, hw, hw, 1], "out_mean_sample": round(chk, 6),
           "parity_max_rel": float(f"{parity:.3g}"), "parity_frames": idx,
           "parity_note": "max |out - oracle| / max |oracle| over the listed frames of the timed call's output; oracle = "
                          "seg_oracle.predict_probs in fp64 on the device, outside the timed region",
           "device_only_ms_per_frame": round(dk * 1e3, 3), "device_tflops": round(91.62e9 / dk / 1e12, 2),
           "device_frac_of_mfma_peak": round(91.62e9 / dk / 1e12 / PEAK, 4)}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def bench_dkl(N=16384, D=2, emit=True):
    """configs[4]: RBF covariance on N=16384 embedded points — HBM-write bound (4*N^2 bytes)."""
    from atomai_amd.nets.gp import kernel_matrix, kernel_matvec, convFeatureExtractor
    rs = np.random.RandomState(0)
    Z = torch.from_numpy(rs.uniform(-1, 1, (N, D)).astype(np.float32)).cuda()
    ls = torch.full((D,), float(np.log(2.0)), device="cuda")
    s2 = float(np.log(2.0))
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        z, l = Z.to(dt), ls.to(dt)
        for _ in range(3): K = kernel_matrix(z, z, l, s2, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K = kernel_matrix(z, z, l, s2, 0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byt = N * N * K.element_size() + N * D * K.element_size()
        res[name] = {"ms": round(ms, 4), "GBps": round(byt / ms / 1e6, 1), "frac_of_8TBps": round(byt / ms / 1e6 / 8000, 4)}
        del K
    v = torch.randn(N, 1, device="cuda")
    for _ in range(2): kernel_matvec(Z, Z, ls, s2, v, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): kernel_matvec(Z, Z, ls, s2, v, 0)
    e1.record(); torch.cuda.synchronize()
    res["matvec_f32_ms"] = round(e0.elapsed_time(e1) / 5, 4)
    if D == 2:
        fe = convFeatureExtractor(256, 2).cuda().eval()
        P = torch.randn(N, 256, device="cuda")
        with torch.no_grad():
            for _ in range(2): fe(P)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): fe(P)
            torch.cuda.synchronize()
        res["conv_extractor_16384x16x16_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    if D == 2:
        # SURVEY section 8-d asks for the embedding dimensions 2 AND 8: D = 8 through the same harness (fp32 / fp64
        # builder + matrix-free product only)
        d8 = bench_dkl(N, 8, emit=False)["detail"]
        res["D8"] = {k: d8[k] for k in ("f32", "f64", "matvec_f32_ms")}
    out = {"metric": f"DKL RBF covariance build, N={N}, D={D}", "value": res["f32"]["ms"], "unit": "ms",
           "higher_is_better": False,
           "roofline": {"bound": "hbm", "achieved": res["f32"]["GBps"], "peak": 8000, "unit": "GB/s",
                        "frac": res["f32"]["frac_of_8TBps"], "traffic": None}, "detail": res}
    if emit:
        print(json.dumps(out), flush=True)
    return out


def bench_dkl_fit(N=16384, patch=16, steps=3, warmup=1, precisions=("single", "double"), emit=True,
                  modes=("kissgp", "exact")):
    """configs[4] END TO END (VERDICT r05 missing #1 / #3): one `dklGPR` training cycle — `dklGPTrainer.train_step`, the unit
    of /root/reference/atomai/trainers/gptrainer.py:126-137 — at N = 16384 flattened 16 x 16 patches with the convolutional
    feature extractor, embedding dimension 2, RBF kernel, for both GP layers of nets/gp.py:
      kissgp (default; the reference's GridInterpolationKernel model, grid 50 x 50): phases = extractor forward, ski_gram
        (weights + W^T W / W^T r, csrc/ski.hip), k_build (K_UU on the 2500 grid nodes), grid_solve / grid_bwd (the dense
        m x m algebra: LU, solves, GEMMs — LIBRARY calls through torch), k_bwd (covariance backward on the grid),
        ski_gram_bwd, extractor backward + glue, Adam;
      exact (dense N x N GP): extractor forward, covariance build, potrf, solve + log-determinant, potri, covariance
        backward, extractor backward + glue, Adam — its O(N^3) library calls dominate.
    Phases by HIP events on the launch stream (`nets.gp.PHASES`); `library_frac` = share of the step in torch.linalg / GEMM."""
    import atomai_amd.nets.gp as gp
    from atomai_amd.nets.gp import convFeatureExtractor
    rs = np.random.RandomState(0)
    X = rs.rand(N, patch * patch).astype(np.float32)
    y = (X.reshape(N, patch, patch)[:, 4:12, 4:12].mean((1, 2)) + 0.05 * rs.randn(N)).astype(np.float32)
    out = {}
    for mode in modes:
      out[mode] = {}
      for prec in precisions:
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        mem0 = torch.cuda.memory_allocated()          # (tensors of earlier bench legs may still be resident: report the delta)
        m = aoi.models.dklGPR(patch * patch, embedim=2, precision=prec, seed=1, gp=mode)
        m.compile_trainer(X, y, training_cycles=1, feature_extractor=convFeatureExtractor)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        per_step, phases = [], {}
        nst = steps if mode == "exact" else max(steps, 10)
        for it in range(warmup + nst):
            gp.PHASES = {} if it >= warmup else None
            e = [ev() for _ in range(5)]
            e[0].record()
            m.optimizer.zero_grad()
            loss = -m.gp_model.mll()
            e[1].record()
            loss.backward()
            e[2].record()
            m.optimizer.step()
            e[3].record()
            lv = loss.item()
            e[4].record(); torch.cuda.synchronize()
            if it >= warmup:
                per_step.append(e[0].elapsed_time(e[4]))
                ph = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in gp.PHASES.items()}
                ph["forward_total"] = e[0].elapsed_time(e[1])
                ph["backward_total"] = e[1].elapsed_time(e[2])
                ph["adam"] = e[2].elapsed_time(e[3])
                for k, v in ph.items():
                    phases.setdefault(k, []).append(v)
            gp.PHASES = None
        med = lambda v: float(np.median(v))
        ph = {k: round(med(v), 3) for k, v in phases.items()}
        step = med(per_step)
        if mode == "exact":
            ph["extractor_bwd_and_glue"] = round(ph["backward_total"] - ph.get("potri", 0.0) - ph.get("k_bwd", 0.0), 3)
            lib = ph.get("potrf", 0.0) + ph.get("potri", 0.0) + ph.get("solve_logdet", 0.0)
        else:
            bw_k = sum(v for k, v in ph.items() if k in ("grid_bwd", "k_bwd", "ski_gram_bwd")) + 0.5 * ph.get("k_build", 0.0)
            ph["extractor_bwd_and_glue"] = round(ph["backward_total"] - bw_k, 3)
            lib = ph.get("grid_solve", 0.0) + ph.get("grid_bwd", 0.0)
        rec = {"ms_per_fit_step": round(step, 2), "phases_ms": ph, "library_ms": round(lib, 2),
               "library_frac": round(lib / step, 4), "optimizer": type(m.optimizer).__name__,
               "peak_mem_GB": round((torch.cuda.max_memory_allocated() - mem0) / 1e9, 2), "loss": round(lv, 6)}
        if mode == "exact":
            with torch.no_grad():
                gpm = m.gp_model
                Z = gpm.embed(gpm.train_inputs[0])
                K = gp.kernel_matrix(Z, Z, gpm.lengthscale[0], float(gpm.outputscale[0]), 0, float(gpm.noise[0, 0]))
                Ki = torch.cholesky_inverse(torch.linalg.cholesky(K))
                asym = float((Ki - Ki.T).abs().max() / Ki.abs().max())
                del K, Ki
            rec.update({"potrf_tflops": round(N ** 3 / 3 / (ph["potrf"] * 1e-3) / 1e12, 2) if ph.get("potrf") else None,
                        "potri_tflops": round(2 * N ** 3 / 3 / (ph["potri"] * 1e-3) / 1e12, 2) if ph.get("potri") else None,
                        "kinv_rel_asymmetry": float(f"{asym:.2e}")})
        else:
            rec["grid_nodes"] = m.gp_model.grid.m
            # the same model state evaluated by the dense exact GP: how far the two layers' objectives are apart
            with torch.no_grad():
                gpm = m.gp_model
                Z = gpm.embed(gpm.train_inputs[0])
                a = gp._SkiMLLFn.apply(Z, gpm.train_targets, gpm.lengthscale, gpm.outputscale, gpm.noise[:, 0],
                                       gpm.mean_constant[:, 0], 0, gpm.grid)
                b = gp._ExactMLLFn.apply(Z, gpm.train_targets[0], gpm.lengthscale[0], gpm.outputscale[0], gpm.noise[0, 0],
                                         gpm.mean_constant[0, 0], 0)
                rec["mll_kissgp_vs_exact"] = [round(float(a), 6), round(float(b), 6)]
        out[mode][prec] = rec
        del m
    first = out[modes[0]][precisions[0]]
    res = {"metric": f"dklGPR fit step ({modes[0]} GP layer, conv extractor, N={N}, {patch}x{patch} patches, embedim 2)",
           "value": first["ms_per_fit_step"], "unit": "ms", "higher_is_better": False, "dtype": precisions[0],
           "note": "kissgp = the reference's GridInterpolationKernel model (grid 50 x 50) with its marginal likelihood evaluated "
                   "exactly through m x m grid algebra (gpytorch: CG / Lanczos estimators); exact = dense N x N GP, whose potrf + "
                   "potri + solve are rocSOLVER / rocBLAS through torch.  library_frac = share of the step inside torch.linalg / "
                   "GEMM calls; the HIP kernels of this build are ski_gram(_bwd), k_build, k_bwd, the extractor and Adam.",
           "detail": out}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def _conv_flops(name, a):
    """Executed MFMA-conv FLOPs of one C-ABI call (stored/padded channel counts, as launched)."""
    if name == "amx_conv2d_fwd":
        return 2.0 * (a[3] + a[7]) * a[19] * a[20] * a[16] * a[17] * a[18]
    if name == "amx_conv2d_fwd_act":
        return 2.0 * (a[4] + a[9]) * a[21] * a[22] * a[18] * a[19] * a[20]
    if name == "amx_conv2d_dgrad":
        return 2.0 * a[1] * (a[5] + a[7]) * a[11] * a[8] * a[9] * a[10]
    if name == "amx_conv2d_dgrad_fused":
        return 2.0 * a[6] * (a[9] + a[11]) * a[15] * a[12] * a[13] * a[14]
    if name == "amx_conv2d_dgrad_fused_bsum":
        return 2.0 * a[6] * a[9] * a[13] * a[10] * a[11] * a[12]
    if name == "amx_conv2d_wgrad_fused":
        return 2.0 * (a[3] + a[7]) * a[20] * a[21] * a[17] * a[18] * a[19]
    if name == "amx_conv2d_wgrad_act":
        return 2.0 * (a[4] + a[9]) * a[22] * a[23] * a[19] * a[20] * a[21]
    return 0.0


def bench_segfamily(models=("SegResNet", "ResHedNet", "dilnet"), hw=512, bs=32, steps=10, warmup=4, emit=True):
    """Training throughput of the other Segmentor families (default widths) at the headline shape."""
    import atomai_amd.engine as eng
    res = {}
    for model in models:
        rs = np.random.RandomState(0)
        ncls = 1 if model == "dilnet" else 3
        X = rs.rand(2 * bs, hw, hw).astype(np.float32)
        y = rs.randint(0, 3, (2 * bs, hw, hw)) if ncls > 1 else (rs.rand(2 * bs, hw, hw) > 0.5).astype(np.float32)
        m = aoi.models.Segmentor(model, nb_classes=ncls, seed=1)
        m.compile_trainer((X, y, X[:bs], y[:bs]), training_cycles=steps + warmup, batch_size=bs,
                          plot_training_history=False)
        flops, orig = [0.0], L.call

        def call(name, *a):
            flops[0] += _conv_flops(name, a)
            return orig(name, *a)
        for i in range(warmup):
            m.train_step(m.X_train[i % 2], m.y_train[i % 2])
        L.call = eng.L.call = call
        m.train_step(m.X_train[0], m.y_train[0])
        L.call = eng.L.call = orig
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            last = m.train_step(m.X_train[i % 2], m.y_train[i % 2])[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        res[model] = {"images_per_s": round(bs / dt, 1), "ms_per_step": round(dt * 1e3, 2), "loss": round(last, 4),
                      "conv_TFLOP_per_step_as_launched": round(flops[0] / 1e12, 3),
                      "step_TFLOPs": round(flops[0] / dt / 1e12, 1), "frac_of_mfma_peak": round(flops[0] / dt / 1e12 / PEAK, 3)}
        del m
        torch.cuda.empty_cache()
    out = {"metric": f"training images/sec, {hw}x{hw}, bs={bs}, default-width families", "unit": "images/s",
           "value": res.get("SegResNet", {}).get("images_per_s"), "detail": res}
    if emit:
        print(json.dumps(out), flush=True)
    return out


def bench_locate(frames=32, hw=1024, C=1, emit=True):
    """Locator on `frames` probability maps of hw x hw (the post-processing of configs[2]): HBM-bound integer
    work; algorithmic bytes = the probabilities read once (4*C bytes per pixel)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import locator_oracle as lo
    from atomai_amd.predictors.locator import locate_device
    rs = np.random.RandomState(0)
    base = lo.synthetic_maps(rs, 2, hw, hw, C, n_blobs=2500, noise=0.3)
    x = torch.from_numpy(np.concatenate([base] * (frames // 2))).cuda()
    recs = timed_calls({"amx_locate_label", "amx_locate_emit"})
    import atomai_amd.predictors.locator as lm
    lm.L.call = L.call
    for _ in range(2):
        d = locate_device(x, 0.5, 5)
    torch.cuda.synchronize(); recs.clear()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        d = locate_device(x, 0.5, 5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t_label = sum(e0.elapsed_time(e1) for n, e0, e1 in recs if n == "amx_locate_label") / reps
    t_emit = sum(e0.elapsed_time(e1) for n, e0, e1 in recs if n == "amx_locate_emit") / reps
    t0 = time.perf_counter()
    ref = lo.locate(base.copy(), 0.5, 5)
    cpu_dt = (time.perf_counter() - t0) / 2
    same = all(np.array_equal(d[i], ref[i]) for i in range(2))
    nbytes = frames * hw * hw * C * 4
    gbps = nbytes / ((t_label + t_emit) * 1e-3) / 1e9
    out = {"metric": f"Locator frames/sec ({hw}x{hw}, {C}-channel maps)", "value": round(frames / dt, 1),
           "unit": "frames/s", "n_gpus": 1, "dtype": "i32", "ms_per_frame_device": round((t_label + t_emit) / frames, 4),
           "ms_label": round(t_label, 3), "ms_emit": round(t_emit, 3), "ms_end_to_end": round(dt * 1e3, 3),
           "centres_per_frame": int(np.mean([len(v) for v in d.values()])), "matches_oracle": bool(same),
           "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000, "unit": "GB/s",
                        "frac": round(gbps / 8000, 4), "traffic": None},
           "cpu_baseline": {"value": round(1 / cpu_dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": "2 frames through oracle/locator_oracle.py (scipy.ndimage)"}}
    if emit:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    what = sys.argv[1:] or ["rvae", "predict"]
    os.makedirs("gpurun_out", exist_ok=True)
    res = {}
    for w in what:
        res[w] = {"rvae": bench_rvae, "predict": bench_predict, "dkl": bench_dkl, "dklfit": bench_dkl_fit, "locate": bench_locate, "segfamily": bench_segfamily,
                  "predict4096": bench_predict_full}[w]()
    json.dump(res, open("gpurun_out/bench_extra.json", "w"), indent=1)
