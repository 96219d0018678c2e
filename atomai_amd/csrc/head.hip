// head.hip — the network head: the final pixel-wise 1x1 convolution `px` (Cout = nb_classes, far too
// narrow for MFMA -> VALU, bandwidth-bound) and the segmentation losses, forward fused with backward.
//
//   self.px = nn.Conv2d(nb_filters, nb_classes, 1, 1, 0)                 atomai/nets/fcnn.py:115, 212
//   select_loss('ce'): CrossEntropyLoss (nb_classes > 2) / BCEWithLogitsLoss (== 1)   atomai/losses_metrics/losses.py:152-155
//   SegPredictor.forward_: softmax(dim=1) / sigmoid, permute to NHWC     atomai/predictors/predictor.py:219-229
#include "amx_device.h"
#ifndef AMX_HEAD_DIV
#define AMX_HEAD_DIV 1          // (image, pixel) of a linear index: 1 = 64-bit division per pixel, 0 = carry-advanced.
                                // Measured in isolation (tools/gpu_small_kernels_ab.py): px_bwd 317 vs 390 us, px_fwd 141
                                // vs 149 us -> the division stays (a 32-bit division behind an `npix < 2^32` test: 461 us)
#endif

#ifndef AMX_PX_BWD_UNROLL
#define AMX_PX_BWD_UNROLL 4     // pixels of a thread in flight in px_bwd (profiles/r03_px_bwd_ab.log: class-templated 307 -> 235 us, 4 in flight 226)
#endif

#define MAXCLS 8

// ------------------------------------------------------------------ px forward
// logits[n][k][h][w] (NCHW, the module's public output) = sum_c xn[p][c] * W[k][c] + b[k],
// xn = a*scale + shift (BN of the previous block applied on load).
// mode 0: raw logits NCHW.  mode 1: probabilities NHWC [P][K] (sigmoid if K == 1 else softmax).
__global__ __launch_bounds__(256) void px_fwd_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ out,
                                                     long npix, long HW, int C, int Cs, int K, int mode) {
    // one thread per pixel: the Cs floats of a pixel are contiguous (a wave reads one contiguous 64*Cs*4 B
    // span), and every class plane is written with unit stride across the wave
    const int G = Cs >> 2;
    const long stride = (long)gridDim.x * 256;
    long n = ((long)blockIdx.x * 256 + threadIdx.x) / HW, hw = ((long)blockIdx.x * 256 + threadIdx.x) - n * HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride, hw += stride) {
#if AMX_HEAD_DIV
        n = p / HW; hw = p - n * HW;
#else
        while (hw >= HW) { hw -= HW; ++n; }                  // (image, pixel) of p without a division per pixel
#endif
        float acc[MAXCLS];
        #pragma unroll
        for (int k = 0; k < MAXCLS; ++k) acc[k] = k < K ? b[k] : 0.f;
        for (int cg = 0; cg < G; ++cg) {
            float4 v = amx_ld4(a + (size_t)p * Cs + cg * 4);
            if (scale) {
                const float4 sc = amx_ld4(scale + cg * 4), sh = amx_ld4(shift + cg * 4);
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            }
            const float vv[4] = {v.x, v.y, v.z, v.w};
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) {
                if (k >= K) break;
                #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cg * 4 + e;
                    if (c < C) acc[k] = fmaf(vv[e], w[k * C + c], acc[k]);
                }
            }
        }
        if (mode == 0) {
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < K) out[((size_t)n * K + k) * HW + hw] = acc[k];
        } else if (K == 1) {
            out[p] = 1.f / (1.f + expf(-acc[0]));
        } else {
            float mx = acc[0];
            #pragma unroll
            for (int k = 1; k < MAXCLS; ++k) if (k < K) mx = fmaxf(mx, acc[k]);
            float s = 0.f;
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < K) { acc[k] = expf(acc[k] - mx); s += acc[k]; }
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < K) out[(size_t)p * K + k] = acc[k] / s;
        }
    }
}

// More than MAXCLS classes (the reference's px / CrossEntropyLoss take any count, fcnn.py:115, losses.py:155): the same
// arithmetic in chunks of MAXCLS classes — a pixel's channels are re-read per chunk (L1 / L2 hits), the per-class FMA
// chain over the channels is the one of the kernel above, so logits agree bit for bit for any K.  mode 1 writes the
// logits of a pixel to its NHWC row first and normalises the row in place (the same thread owns it).
__global__ __launch_bounds__(256) void px_fwd_generic_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ out,
                                                             long npix, long HW, int C, int Cs, int K, int mode) {
    const int G = Cs >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
        const long n = p / HW, hw = p - n * HW;
        for (int k0 = 0; k0 < K; k0 += MAXCLS) {
            const int kc = K - k0 < MAXCLS ? K - k0 : MAXCLS;
            float acc[MAXCLS];
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) acc[k] = k < kc ? b[k0 + k] : 0.f;
            for (int cg = 0; cg < G; ++cg) {
                float4 v = amx_ld4(a + (size_t)p * Cs + cg * 4);
                if (scale) {
                    const float4 sc = amx_ld4(scale + cg * 4), sh = amx_ld4(shift + cg * 4);
                    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                    v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                }
                const float vv[4] = {v.x, v.y, v.z, v.w};
                #pragma unroll
                for (int k = 0; k < MAXCLS; ++k) {
                    if (k >= kc) break;
                    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = cg * 4 + e;
                        if (c < C) acc[k] = fmaf(vv[e], w[(k0 + k) * C + c], acc[k]);
                    }
                }
            }
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) {
                if (k >= kc) break;
                if (mode == 0) out[((size_t)n * K + k0 + k) * HW + hw] = acc[k];
                else out[(size_t)p * K + k0 + k] = acc[k];
            }
        }
        if (mode != 0) {                                        // softmax of the row this thread has just written
            float* row = out + (size_t)p * K;
            float mx = row[0];
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, row[k]);
            float s = 0.f;
            for (int k = 0; k < K; ++k) { const float e = expf(row[k] - mx); row[k] = e; s += e; }
            for (int k = 0; k < K; ++k) row[k] = row[k] / s;
        }
    }
}

extern "C" int amx_px_fwd(const float* a, const float* scale, const float* shift, const float* w,
                          const float* b, float* out, int N, int H, int W, int C, int Cs, int K, int mode,
                          void* stream) {
    if (!a || !w || !b || !out || (Cs & 3) || C <= 0 || Cs < C || Cs > 256) AMX_BADARG(1);
    if (K < 1) AMX_BADARG(2);
    if (K > MAXCLS) {
        if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
        const long npix_g = (long)N * H * W;
        long nbg = (npix_g + 255) / 256;
        if (nbg > 16384) nbg = 16384;
        AMX_LAUNCH(px_fwd_generic_kernel, dim3((unsigned)nbg), dim3(256), 0, (hipStream_t)stream, a, scale, shift, w, b,
                   out, npix_g, (long)H * W, C, Cs, K, mode);
        AMX_CHECK_LAUNCH();
        return 0;
    }
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
    const long npix = (long)N * H * W;
    long nb = (npix + 255) / 256;
    if (nb > 16384) nb = 16384;
    AMX_LAUNCH(px_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, scale, shift, w, b,
               out, npix, (long)H * W, C, Cs, K, mode);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ px backward
// dxn[p][c] = sum_k dl[n][k][hw] * W[k][c];  partial rows: part[blk][K][Cs] (dW) and partb[blk][K] (db)
// KT: compile-time bound of the class loops (K itself for K <= 4 — the reference's nets have 1-3 classes — else MAXCLS):
// with the loops bounded by MAXCLS the per-class registers of 8 classes were live (122 VGPRs at one pixel in flight).
template <int KT>
__global__ __launch_bounds__(256) void px_bwd_kernel(const float* __restrict__ dl,
                                                     const float* __restrict__ a,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift,
                                                     const float* __restrict__ w, float* __restrict__ dxn,
                                                     float* __restrict__ part, float* __restrict__ partb,
                                                     float* __restrict__ bstats,
                                                     long npix, long HW, int C, int Cs, int K, int ppb,
                                                     int Kall, int kbase, int accum) {
    // (K classes kbase .. kbase + K - 1 of Kall: one launch when Kall <= MAXCLS; more classes run as chunks of MAXCLS,
    //  every chunk after the first adding to the dxn the earlier ones wrote — `accum` — and only the last one, which
    //  sees the final dxn, forming the BatchNorm-backward sums `bstats`)
    const int G = Cs >> 2, PL = 256 / G;
    const int tid = threadIdx.x;
    const int pl = tid / G, cg = tid - pl * G;
    const bool active = pl < PL;
    AMX_DYN_SMEM(float, s);                       // [PL][K][Cs] + [PL][K] + [2][PL][Cs]
    float4 bs1 = make_float4(0, 0, 0, 0), bs2 = make_float4(0, 0, 0, 0);   // sum dxn, sum dxn * a (raw)
    float4 dw[KT];
    float db[KT];
    #pragma unroll
    for (int k = 0; k < KT; ++k) { dw[k] = make_float4(0, 0, 0, 0); db[k] = 0.f; }
    float4 wk[KT];
    #pragma unroll
    for (int k = 0; k < KT; ++k) {
        wk[k] = make_float4(0, 0, 0, 0);
        if (k < K && active) {
            const int c = cg * 4;
            const float* wr = w + (size_t)(kbase + k) * C;
            wk[k].x = c + 0 < C ? wr[c + 0] : 0.f; wk[k].y = c + 1 < C ? wr[c + 1] : 0.f;
            wk[k].z = c + 2 < C ? wr[c + 2] : 0.f; wk[k].w = c + 3 < C ? wr[c + 3] : 0.f;
        }
    }
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    if (active) {
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (scale) { sc = amx_ld4(scale + cg * 4); sh = amx_ld4(shift + cg * 4); }
        // U pixels of this thread in flight: every load of the U pixels (a: 16 B, dl: K scalars) is issued before the
        // first FMA; the sums are still formed in pixel order.  With one pixel per iteration the loop was a chain of
        // ppb / PL global-load round trips per thread (3.7 TB/s).  Loads are unconditional (clamped to the thread's
        // first pixel of the iteration), see conv1.hip.
        constexpr int U = AMX_PX_BWD_UNROLL;
        for (long p = p0 + pl; p < p1; p += (long)PL * U) {
            float4 av[U];
            float gv[U][KT];
            bool ok[U];
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = p + (long)u * PL < p1;
                const long pu = ok[u] ? p + (long)u * PL : p;
                const long n = pu / HW, hw = pu - n * HW;
                av[u] = amx_ld4(a + (size_t)pu * Cs + cg * 4);
                #pragma unroll
                for (int k = 0; k < KT; ++k) gv[u][k] = k < K ? dl[((size_t)n * Kall + kbase + k) * HW + hw] : 0.f;
            }
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                float4 v = av[u];
                const float4 raw = v;
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                float4 d = make_float4(0, 0, 0, 0);
                if (accum) d = amx_ld4(dxn + (size_t)(p + (long)u * PL) * Cs + cg * 4);
                #pragma unroll
                for (int k = 0; k < KT; ++k) {
                    if (k >= K) break;
                    const float g = gv[u][k];
                    d.x = fmaf(g, wk[k].x, d.x); d.y = fmaf(g, wk[k].y, d.y);
                    d.z = fmaf(g, wk[k].z, d.z); d.w = fmaf(g, wk[k].w, d.w);
                    dw[k].x = fmaf(g, v.x, dw[k].x); dw[k].y = fmaf(g, v.y, dw[k].y);
                    dw[k].z = fmaf(g, v.z, dw[k].z); dw[k].w = fmaf(g, v.w, dw[k].w);
                    db[k] += g;
                }
                amx_st4(dxn + (size_t)(p + (long)u * PL) * Cs + cg * 4, d);
                bs1.x += d.x; bs1.y += d.y; bs1.z += d.z; bs1.w += d.w;
                bs2.x = fmaf(d.x, raw.x, bs2.x); bs2.y = fmaf(d.y, raw.y, bs2.y);
                bs2.z = fmaf(d.z, raw.z, bs2.z); bs2.w = fmaf(d.w, raw.w, bs2.w);
            }
        }
        #pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (k >= K) break;
            amx_st4(s + ((size_t)(pl * K + k) * Cs + cg * 4), dw[k]);
            if (cg == 0) s[(size_t)PL * K * Cs + pl * K + k] = db[k];
        }
        float* sb = s + (size_t)PL * K * Cs + (size_t)PL * K;
        sb = sb + ((4 - ((size_t)(sb - s) & 3)) & 3);            // keep float4 alignment
        amx_st4(sb + ((size_t)pl * Cs + cg * 4), bs1);
        amx_st4(sb + ((size_t)(PL + pl) * Cs + cg * 4), bs2);
    }
    __syncthreads();
    for (int i = tid; i < K * Cs; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < PL; ++q) acc += s[(size_t)q * K * Cs + i];
        part[((size_t)blockIdx.x * Kall + kbase) * Cs + i] = acc;
    }
    if (tid < K) {
        float acc = 0.f;
        for (int q = 0; q < PL; ++q) acc += s[(size_t)PL * K * Cs + q * K + tid];
        partb[(size_t)blockIdx.x * Kall + kbase + tid] = acc;
    }
    if (bstats) {
        const float* sb = s + (size_t)PL * K * Cs + (size_t)PL * K;
        sb = sb + ((4 - ((size_t)(sb - s) & 3)) & 3);
        for (int i = tid; i < 2 * Cs; i += 256) {
            const int which = i / Cs, c = i - which * Cs;
            float acc = 0.f;
            for (int q = 0; q < PL; ++q) acc += sb[(size_t)(which * PL + q) * Cs + c];
            bstats[((size_t)blockIdx.x * 2 + which) * Cs + c] = acc;
        }
    }
}

extern "C" int amx_px_bwd(const float* dl, const float* a, const float* scale, const float* shift,
                          const float* w, float* dxn, float* part, float* partb, float* bstats, int N, int H,
                          int W, int C, int Cs, int K, int rows, int rows_pix, void* stream) {
    if (!dl || !a || !w || !dxn || !part || !partb || (Cs & 3) || C <= 0 || Cs < C || Cs > 256)
        AMX_BADARG(1);
    if (K < 1) AMX_BADARG(2);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
    const long npix = (long)N * H * W;
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(4);
    const int PL = 256 / (Cs / 4);
    if (K > MAXCLS) {                                           // chunks of MAXCLS classes (see the kernel's note)
        for (int k0 = 0; k0 < K; k0 += MAXCLS) {
            const int kc = K - k0 < MAXCLS ? K - k0 : MAXCLS;
            const size_t ldc = ((size_t)PL * kc * Cs + (size_t)PL * kc + 4 + (size_t)2 * PL * Cs) * sizeof(float);
            AMX_LAUNCH(px_bwd_kernel<MAXCLS>, dim3(rows), dim3(256), ldc, (hipStream_t)stream, dl, a, scale, shift, w, dxn,
                       part, partb, (k0 + kc == K) ? bstats : (float*)nullptr, npix, (long)H * W, C, Cs, kc, rows_pix, K,
                       k0, k0 > 0 ? 1 : 0);
            AMX_CHECK_LAUNCH();
        }
        return 0;
    }
    const size_t lds = ((size_t)PL * K * Cs + (size_t)PL * K + 4 + (size_t)2 * PL * Cs) * sizeof(float);
#define PX_BWD_LAUNCH(KT_)                                                                                              \
    AMX_LAUNCH(px_bwd_kernel<KT_>, dim3(rows), dim3(256), lds, (hipStream_t)stream, dl, a, scale, shift, w, dxn, part, \
               partb, bstats, npix, (long)H * W, C, Cs, K, rows_pix, K, 0, 0)
    switch (K) {
        case 1: PX_BWD_LAUNCH(1); break;
        case 2: PX_BWD_LAUNCH(2); break;
        case 3: PX_BWD_LAUNCH(3); break;
        case 4: PX_BWD_LAUNCH(4); break;
        default: PX_BWD_LAUNCH(MAXCLS); break;
    }
#undef PX_BWD_LAUNCH
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ px forward + cross entropy + px backward in ONE pass
// The training step of a Segmentor ends  logits = px(xn);  loss = CrossEntropyLoss()(logits, target);  loss.backward()
// (atomai/trainers/trainer.py:201-207).  As three kernels that is px_fwd (reads the last activation, writes the logits),
// ce_fwd_bwd (reads the logits, writes their gradient) and px_bwd (reads the activation AGAIN and the logits gradient, writes
// the activation's gradient): 1.9 GB of traffic at bs 32 / 512^2 / 16 channels, all on the dependency chain between forward
// and backward where nothing else can run.  Here one pass over the activation forms a pixel's logits in registers (the G
// lanes that share a pixel in px_bwd's layout add their partial dot products by xor-shuffles), its softmax / loss term, the
// logits gradient (softmax - onehot) / npix, and from it everything px_bwd produces: dxn, the px weight / bias gradient rows
// and the BatchNorm-backward sums of the producing layer.  Neither the logits nor their gradient exist in memory.
// Called by the trainers' fused step only (nets/fcnn.py: forward_loss); K <= 4 classes (K = 1: BCE), Cs / 4 a power of two <= 64.
// lpart [rows]: per-workgroup loss sums (amx_reduce_rows with 1 / npix gives the mean, as for amx_ce_fwd_bwd).
// BCE (KT == 1): the one-class head with BCEWithLogitsLoss against a float mask `tgtf` (select_loss('ce', 1), the
// reference's default nb_classes): loss and gradient as bce_fwd_bwd_kernel forms them.
template <int KT, bool BCE = false>
__global__ __launch_bounds__(256) void px_ce_train_kernel(const float* __restrict__ a, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ b, const long long* __restrict__ tgt,
                                                          const float* __restrict__ tgtf,
                                                          float* __restrict__ dxn, float* __restrict__ part,
                                                          float* __restrict__ partb, float* __restrict__ bstats,
                                                          float* __restrict__ lpart, long npix, int C, int Cs, int ppb,
                                                          float inv_count) {
    constexpr int K = KT;
    const int G = Cs >> 2, PL = 256 / G;                 // (256 % G == 0: every thread is active)
    const int tid = threadIdx.x;
    const int pl = tid / G, cg = tid - pl * G;
    AMX_DYN_SMEM(float, s);                       // [PL][K][Cs] + [PL][K] (+ pad) + [2][PL][Cs] + [256]
    float4 bs1 = make_float4(0, 0, 0, 0), bs2 = make_float4(0, 0, 0, 0);   // sum dxn, sum dxn * a (raw)
    float4 dw[KT];
    float db[KT], bk[KT];
    float4 wk[KT];
    #pragma unroll
    for (int k = 0; k < KT; ++k) {
        dw[k] = make_float4(0, 0, 0, 0); db[k] = 0.f; bk[k] = b[k];
        const int c = cg * 4;
        const float* wr = w + (size_t)k * C;
        wk[k].x = c + 0 < C ? wr[c + 0] : 0.f; wk[k].y = c + 1 < C ? wr[c + 1] : 0.f;
        wk[k].z = c + 2 < C ? wr[c + 2] : 0.f; wk[k].w = c + 3 < C ? wr[c + 3] : 0.f;
    }
    float lsum = 0.f;
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
    if (scale) { sc = amx_ld4(scale + cg * 4); sh = amx_ld4(shift + cg * 4); }
    constexpr int U = AMX_PX_BWD_UNROLL;
    for (long pb = p0; pb < p1; pb += (long)PL * U) {       // (workgroup-uniform trip count: the shuffles need whole waves)
        const long p = pb + pl;
        float4 av[U];
        int tv[U];
        float tf[U];
        bool ok[U];
        #pragma unroll
        for (int u = 0; u < U; ++u) {             // every load of the U pixels before the first use (see px_bwd_kernel)
            ok[u] = p + (long)u * PL < p1;
            const long pu = ok[u] ? p + (long)u * PL : p0;
            av[u] = amx_ld4(a + (size_t)pu * Cs + cg * 4);
            if (BCE) { tf[u] = tgtf[pu]; tv[u] = 0; } else { tv[u] = (int)tgt[pu]; tf[u] = 0.f; }
        }
        #pragma unroll
        for (int u = 0; u < U; ++u) {
            // (no `continue` for a missing pixel: the shuffles below need every lane of the wave)
            float4 v = av[u];
            const float4 raw = v;
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            float lg[KT];
            #pragma unroll
            for (int k = 0; k < KT; ++k) {
                float t = v.x * wk[k].x;
                t = fmaf(v.y, wk[k].y, t); t = fmaf(v.z, wk[k].z, t); t = fmaf(v.w, wk[k].w, t);
                for (int o = 1; o < G; o <<= 1) t += __shfl_xor(t, o);
                lg[k] = t + bk[k];
            }
            float gk[KT], lterm;
            if (BCE) {
                const float xv = lg[0], e = expf(-fabsf(xv));
                lterm = fmaxf(xv, 0.f) - xv * tf[u] + log1pf(e);
                gk[0] = ((xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e)) - tf[u]) * inv_count;
            } else {
                float mx = lg[0];
                #pragma unroll
                for (int k = 1; k < KT; ++k) mx = fmaxf(mx, lg[k]);
                float se = 0.f, xt = 0.f;
                #pragma unroll
                for (int k = 0; k < KT; ++k) {
                    if (k == tv[u]) xt = lg[k] - mx;
                    lg[k] = expf(lg[k] - mx); se += lg[k];
                }
                const float inv_s = 1.f / se;
                lterm = logf(se) - xt;
                #pragma unroll
                for (int k = 0; k < KT; ++k) gk[k] = (lg[k] * inv_s - (k == tv[u] ? 1.f : 0.f)) * inv_count;
            }
            if (!ok[u]) continue;
            if (cg == 0) lsum += lterm;
            float4 d = make_float4(0, 0, 0, 0);
            #pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float g = gk[k];
                d.x = fmaf(g, wk[k].x, d.x); d.y = fmaf(g, wk[k].y, d.y);
                d.z = fmaf(g, wk[k].z, d.z); d.w = fmaf(g, wk[k].w, d.w);
                dw[k].x = fmaf(g, v.x, dw[k].x); dw[k].y = fmaf(g, v.y, dw[k].y);
                dw[k].z = fmaf(g, v.z, dw[k].z); dw[k].w = fmaf(g, v.w, dw[k].w);
                db[k] += g;
            }
            amx_st4(dxn + (size_t)(p + (long)u * PL) * Cs + cg * 4, d);
            bs1.x += d.x; bs1.y += d.y; bs1.z += d.z; bs1.w += d.w;
            bs2.x = fmaf(d.x, raw.x, bs2.x); bs2.y = fmaf(d.y, raw.y, bs2.y);
            bs2.z = fmaf(d.z, raw.z, bs2.z); bs2.w = fmaf(d.w, raw.w, bs2.w);
        }
    }
    #pragma unroll
    for (int k = 0; k < KT; ++k) {
        amx_st4(s + ((size_t)(pl * K + k) * Cs + cg * 4), dw[k]);
        if (cg == 0) s[(size_t)PL * K * Cs + pl * K + k] = db[k];
    }
    float* sb = s + (size_t)PL * K * Cs + (size_t)PL * K;
    sb = sb + ((4 - ((size_t)(sb - s) & 3)) & 3);            // keep float4 alignment
    amx_st4(sb + ((size_t)pl * Cs + cg * 4), bs1);
    amx_st4(sb + ((size_t)(PL + pl) * Cs + cg * 4), bs2);
    float* red = sb + (size_t)2 * PL * Cs;
    red[tid] = lsum;
    __syncthreads();
    for (int i = tid; i < K * Cs; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < PL; ++q) acc += s[(size_t)q * K * Cs + i];
        part[(size_t)blockIdx.x * K * Cs + i] = acc;
    }
    if (tid < K) {
        float acc = 0.f;
        for (int q = 0; q < PL; ++q) acc += s[(size_t)PL * K * Cs + q * K + tid];
        partb[(size_t)blockIdx.x * K + tid] = acc;
    }
    if (bstats) {
        for (int i = tid; i < 2 * Cs; i += 256) {
            const int which = i / Cs, c = i - which * Cs;
            float acc = 0.f;
            for (int q = 0; q < PL; ++q) acc += sb[(size_t)(which * PL + q) * Cs + c];
            bstats[((size_t)blockIdx.x * 2 + which) * Cs + c] = acc;
        }
    }
    for (int o = 128; o > 0; o >>= 1) { __syncthreads(); if (tid < o) red[tid] += red[tid + o]; }
    if (tid == 0) lpart[blockIdx.x] = red[0];
}

extern "C" int amx_px_ce_train_supported(int Cs, int K) {
    const int G = Cs >> 2;
    return (Cs > 0 && !(Cs & 3) && Cs <= 256 && (G & (G - 1)) == 0 && K >= 1 && K <= 4) ? 1 : 0;
}

extern "C" int amx_px_ce_train(const float* a, const float* scale, const float* shift, const float* w, const float* b,
                               const long long* target, const float* target_f, float* dxn, float* part, float* partb,
                               float* bstats, float* lpart, int N, int H, int W, int C, int Cs, int K, int rows,
                               int rows_pix, void* stream) {
    if (!a || !w || !b || !dxn || !part || !partb || !lpart || C <= 0 || Cs < C) AMX_BADARG(1);
    if (!amx_px_ce_train_supported(Cs, K) || (K == 1 ? !target_f : !target)) AMX_BADARG(2);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
    const long npix = (long)N * H * W;
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(4);
    const int PL = 256 / (Cs / 4);
    const size_t lds = ((size_t)PL * K * Cs + (size_t)PL * K + 4 + (size_t)2 * PL * Cs + 256) * sizeof(float);
#define PX_CE_LAUNCH(KT_)                                                                                             \
    AMX_LAUNCH((px_ce_train_kernel<KT_, KT_ == 1>), dim3(rows), dim3(256), lds, (hipStream_t)stream, a, scale, shift, w, b, \
               target, target_f, dxn, part, partb, bstats, lpart, npix, C, Cs, rows_pix, 1.0f / (float)npix)
    switch (K) {
        case 1: PX_CE_LAUNCH(1); break;
        case 2: PX_CE_LAUNCH(2); break;
        case 3: PX_CE_LAUNCH(3); break;
        default: PX_CE_LAUNCH(4); break;
    }
#undef PX_CE_LAUNCH
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ cross entropy (fwd + bwd fused)
// logits NCHW [N][K][HW], target int64 [N][HW].  loss = mean over valid pixels of (lse - x_t);
// dlogits = (softmax - onehot) / n_valid is written in the same pass (NCHW) so that backward is free.
// part[blk][2] = (sum of per-pixel losses, number of valid pixels); the mean is taken in the finalize
// kernel in fp64 and the gradient is scaled by 1/n_valid there-after by the caller-provided inv_count
// (n_valid is known a priori when no ignore_index is present: the reference never produces one).
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ x,
                                                         const long long* __restrict__ tgt,
                                                         float* __restrict__ dx, float* __restrict__ part,
                                                         long npix, long HW, int K, float inv_count) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float lsum = 0.f;
    const long stride = (long)gridDim.x * 256;
    long n = ((long)blockIdx.x * 256 + tid) / HW, hw = ((long)blockIdx.x * 256 + tid) - n * HW;
    for (long p = (long)blockIdx.x * 256 + tid; p < npix; p += stride, hw += stride) {
#if AMX_HEAD_DIV
        n = p / HW; hw = p - n * HW;
#else
        while (hw >= HW) { hw -= HW; ++n; }                  // (no 64-bit division per pixel)
#endif
        const float* xp = x + (size_t)n * K * HW + hw;
        float v[MAXCLS];
        float mx = -3.4e38f;
        #pragma unroll
        for (int k = 0; k < MAXCLS; ++k) { if (k >= K) break; v[k] = xp[(size_t)k * HW]; mx = fmaxf(mx, v[k]); }
        const int t = (int)tgt[p];
        float s = 0.f, xt = 0.f;
        #pragma unroll
        for (int k = 0; k < MAXCLS; ++k) {
            if (k >= K) break;
            if (k == t) xt = v[k] - mx;
            v[k] = expf(v[k] - mx); s += v[k];
        }
        const float inv_s = 1.f / s;
        #pragma unroll
        for (int k = 0; k < MAXCLS; ++k) {
            if (k >= K) break;
            const float sm = v[k] * inv_s;
            if (dx) dx[(size_t)n * K * HW + (size_t)k * HW + hw] = (sm - (k == t ? 1.f : 0.f)) * inv_count;
        }
        lsum += logf(s) - xt;
    }
    red[tid] = lsum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) part[blockIdx.x] = red[0];
}

// More than MAXCLS classes: the logits of a pixel are read three times (max, sum of exponentials, gradient) instead of
// being kept in registers; same operations in the same order as above.
__global__ __launch_bounds__(256) void ce_fwd_bwd_generic_kernel(const float* __restrict__ x,
                                                                 const long long* __restrict__ tgt,
                                                                 float* __restrict__ dx, float* __restrict__ part,
                                                                 long npix, long HW, int K, float inv_count) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float lsum = 0.f;
    for (long p = (long)blockIdx.x * 256 + tid; p < npix; p += (long)gridDim.x * 256) {
        const long n = p / HW, hw = p - n * HW;
        const float* xp = x + (size_t)n * K * HW + hw;
        float mx = -3.4e38f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, xp[(size_t)k * HW]);
        const int t = (int)tgt[p];
        float s = 0.f, xt = 0.f;
        for (int k = 0; k < K; ++k) {
            const float v = xp[(size_t)k * HW] - mx;
            if (k == t) xt = v;
            s += expf(v);
        }
        if (dx) {
            const float inv_s = 1.f / s;
            for (int k = 0; k < K; ++k) {
                const float sm = expf(xp[(size_t)k * HW] - mx) * inv_s;
                dx[(size_t)n * K * HW + (size_t)k * HW + hw] = (sm - (k == t ? 1.f : 0.f)) * inv_count;
            }
        }
        lsum += logf(s) - xt;
    }
    red[tid] = lsum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) part[blockIdx.x] = red[0];
}

extern "C" int amx_ce_fwd_bwd(const float* logits, const long long* target, float* dlogits, float* part,
                              int rows, int N, int K, long HW, void* stream) {
    if (!logits || !target || !part || K < 2 || rows <= 0) AMX_BADARG(1);
    const long npix = (long)N * HW;
    if (K > MAXCLS) {
        AMX_LAUNCH(ce_fwd_bwd_generic_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, target, dlogits,
                   part, npix, HW, K, 1.0f / (float)npix);
        AMX_CHECK_LAUNCH();
        return 0;
    }
    AMX_LAUNCH(ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, target, dlogits,
               part, npix, HW, K, 1.0f / (float)npix);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ BCE with logits (fwd + bwd fused)
// loss = mean( max(x,0) - x*y + log(1 + exp(-|x|)) );  dx = (sigmoid(x) - y) / numel
__global__ __launch_bounds__(256) void bce_fwd_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          float* __restrict__ dx, float* __restrict__ part,
                                                          long n, float inv_count) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float lsum = 0.f;
    for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i], t = y[i];
        const float e = expf(-fabsf(v));
        lsum += fmaxf(v, 0.f) - v * t + log1pf(e);
        if (dx) {
            const float sg = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            dx[i] = (sg - t) * inv_count;
        }
    }
    red[tid] = lsum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) part[blockIdx.x] = red[0];
}

extern "C" int amx_bce_fwd_bwd(const float* logits, const float* target, float* dlogits, float* part,
                               int rows, long numel, void* stream) {
    if (!logits || !target || !part || rows <= 0 || numel <= 0) AMX_BADARG(1);
    AMX_LAUNCH(bce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, target, dlogits,
               part, numel, 1.0f / (float)numel);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ upstream gradient of the scalar loss
__global__ __launch_bounds__(256) void scale_unless_one_kernel(float* __restrict__ x, const float* __restrict__ g, long n) {
    const float f = *g;
    if (f == 1.0f) return;                       // (uniform) the usual case: nothing is read or written
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= f;
}

extern "C" int amx_scale_unless_one(float* x, const float* g, long n, void* stream) {
    if (!x || !g || n <= 0) AMX_BADARG(1);
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    AMX_LAUNCH(scale_unless_one_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, g, n);
    AMX_CHECK_LAUNCH();
    return 0;
}

// amx_scale_unless_one for up to four tensors in one launch (the fused training head hands dxn, its two gradient-row
// tensors and the BatchNorm-backward sums to backward at once)
struct ScaleSegs { float* x[4]; long n[4]; long start[5]; int nseg; };
__global__ __launch_bounds__(256) void scale_unless_one_multi_kernel(ScaleSegs sg, const float* __restrict__ g) {
    const float f = *g;
    if (f == 1.0f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < sg.start[sg.nseg]; i += (long)gridDim.x * 256) {
        int k = 0;
        while (i >= sg.start[k + 1]) ++k;
        sg.x[k][i - sg.start[k]] *= f;
    }
}

extern "C" int amx_scale_unless_one_multi(float* x0, long n0, float* x1, long n1, float* x2, long n2, float* x3, long n3,
                                          const float* g, void* stream) {
    if (!g || !x0 || n0 <= 0) AMX_BADARG(1);
    ScaleSegs sg = {};
    float* xs[4] = {x0, x1, x2, x3};
    const long ns[4] = {n0, n1, n2, n3};
    for (int k = 0; k < 4; ++k) {
        if (!xs[k] || ns[k] <= 0) continue;
        sg.x[sg.nseg] = xs[k]; sg.n[sg.nseg] = ns[k]; sg.start[sg.nseg + 1] = sg.start[sg.nseg] + ns[k]; ++sg.nseg;
    }
    long nb = (sg.start[sg.nseg] + 255) / 256;
    if (nb > 4096) nb = 4096;
    AMX_LAUNCH(scale_unless_one_multi_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, sg, g);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ IoU confusion counts (SegTrainer.accuracy_fn)
// The reference's IoU (losses_metrics/metrics.py:16-95) moves logits and labels to the host, thresholds the softmax /
// sigmoid probabilities at `thresh` (cv2.threshold THRESH_BINARY: p > thresh -> 1), squeezes the channels into a class
// map (sum_c c * [p_c > thresh], values above K-1 clipped to 0) and builds a K x K confusion histogram per image with
// torch.bincount.  Here one pass over the logits produces the same per-image integer histograms on the device (integer
// atomics: exact and order independent); the host reads N*Kc*Kc integers and finishes the 10-flop Jaccard arithmetic.
// logits NCHW [N][K][HW]; truth int64 [N][HW] (K > 1) or float [N][HW] (K == 1, binary masks); Kc = max(K, 2);
// hist int32 [N][Kc][Kc] (zeroed by the caller).  Pixels whose label is outside [0, Kc) are skipped (metrics.py:73).
// activation: 1 = x holds logits (softmax / sigmoid applied here, metrics.py:37-41), 0 = x holds probabilities already.
__global__ __launch_bounds__(256) void iou_hist_kernel(const float* __restrict__ x, const long long* __restrict__ ti,
                                                       const float* __restrict__ tf, int* __restrict__ hist,
                                                       long HW, int K, int Kc, float thresh, int blocks_per_img,
                                                       int activation) {
    __shared__ int sh[MAXCLS * MAXCLS < 4 ? 4 : MAXCLS * MAXCLS];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / blocks_per_img, b = blockIdx.x - n * blocks_per_img;
    for (int i = tid; i < Kc * Kc; i += 256) sh[i] = 0;
    __syncthreads();
    for (long hw = (long)b * 256 + tid; hw < HW; hw += (long)blocks_per_img * 256) {
        const float* xp = x + (size_t)n * K * HW + hw;
        int pred = 0;
        if (K == 1) {
            const float p = activation ? 1.f / (1.f + expf(-xp[0])) : xp[0];
            pred = p > thresh ? 1 : 0;
        } else {
            float v[MAXCLS];
            float mx = -3.4e38f;
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) { if (k >= K) break; v[k] = xp[(size_t)k * HW]; mx = fmaxf(mx, v[k]); }
            float s = 1.f;
            if (activation) {
                s = 0.f;
                #pragma unroll
                for (int k = 0; k < MAXCLS; ++k) { if (k >= K) break; v[k] = expf(v[k] - mx); s += v[k]; }
            }
            #pragma unroll
            for (int k = 0; k < MAXCLS; ++k) { if (k >= K) break; if (v[k] / s > thresh) pred += k; }
            if (pred > K - 1) pred = 0;                          // squeeze_channels(clip=True), imaug.py:385-386
        }
        const long t = ti ? (long)ti[(size_t)n * HW + hw] : (long)tf[(size_t)n * HW + hw];   // .long(): truncation
        if (t >= 0 && t < Kc) atomicAdd(&sh[(int)t * Kc + pred], 1);
    }
    __syncthreads();
    for (int i = tid; i < Kc * Kc; i += 256)
        if (sh[i]) atomicAdd(&hist[(size_t)n * Kc * Kc + i], sh[i]);
}

// More than MAXCLS classes: the class scores of a pixel are re-read instead of held in registers, and the K x K counts
// go to the global histogram directly (integer atomics: still exact).
__global__ __launch_bounds__(256) void iou_hist_generic_kernel(const float* __restrict__ x, const long long* __restrict__ ti,
                                                               const float* __restrict__ tf, int* __restrict__ hist,
                                                               long HW, int K, float thresh, int blocks_per_img,
                                                               int activation) {
    // K <= 32: the block's K x K counts are privatised in LDS (integer atomics: exact, order independent) and reach the
    // global histogram once per block and bin — one global atomic per PIXEL contended on a few hot bins
    __shared__ int s_hist[32 * 32];
    const int tid = threadIdx.x;
    const bool priv = K <= 32;
    if (priv) { for (int i = tid; i < K * K; i += 256) s_hist[i] = 0; __syncthreads(); }
    const int n = blockIdx.x / blocks_per_img, b = blockIdx.x - n * blocks_per_img;
    for (long hw = (long)b * 256 + tid; hw < HW; hw += (long)blocks_per_img * 256) {
        const float* xp = x + (size_t)n * K * HW + hw;
        float mx = -3.4e38f, s = 1.f;
        if (activation) {
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, xp[(size_t)k * HW]);
            s = 0.f;
            for (int k = 0; k < K; ++k) s += expf(xp[(size_t)k * HW] - mx);
        }
        int pred = 0;
        for (int k = 0; k < K; ++k) {
            const float v = activation ? expf(xp[(size_t)k * HW] - mx) : xp[(size_t)k * HW];
            if (v / s > thresh) pred += k;
        }
        if (pred > K - 1) pred = 0;
        const long t = ti ? (long)ti[(size_t)n * HW + hw] : (long)tf[(size_t)n * HW + hw];
        if (t >= 0 && t < K) {
            if (priv) atomicAdd(&s_hist[(int)t * K + pred], 1);
            else atomicAdd(&hist[((size_t)n * K + (int)t) * K + pred], 1);
        }
    }
    if (priv) {
        __syncthreads();
        for (int i = tid; i < K * K; i += 256)
            if (s_hist[i]) atomicAdd(&hist[(size_t)n * K * K + i], s_hist[i]);
    }
}

extern "C" int amx_iou_hist(const float* logits, const long long* truth_i64, const float* truth_f32, int N, int K,
                            long HW, float thresh, int activation, int* hist, void* stream) {
    if (!logits || !hist || N <= 0 || HW <= 0 || K < 1) AMX_BADARG(1);
    if ((truth_i64 == nullptr) == (truth_f32 == nullptr)) AMX_BADARG(2);
    const int Kc = K < 2 ? 2 : K;
    long bpi = (HW + 256L * 16 - 1) / (256L * 16);               // ~16 pixels per thread
    if (bpi < 1) bpi = 1;
    if (bpi > 1024) bpi = 1024;
    if (K > MAXCLS)
        AMX_LAUNCH(iou_hist_generic_kernel, dim3((unsigned)(N * bpi)), dim3(256), 0, (hipStream_t)stream, logits,
                   truth_i64, truth_f32, hist, HW, K, thresh, (int)bpi, activation ? 1 : 0);
    else
        AMX_LAUNCH(iou_hist_kernel, dim3((unsigned)(N * bpi)), dim3(256), 0, (hipStream_t)stream, logits, truth_i64,
                   truth_f32, hist, HW, K, Kc, thresh, (int)bpi, activation ? 1 : 0);
    AMX_CHECK_LAUNCH();
    return 0;
}
