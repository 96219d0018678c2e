// wgrad_ws.hip — wave-SPECIALISED weight gradient of the plain 3x3 convolutions (round 4; autograd of nn.Conv2d in
// atomai/nets/blocks.py:59-76, fcnn.py:100-138 — the reference gets it from ATen's conv backward).
//
// Why a second kernel: wgrad_kernel.h runs ONE persistent 4-wave workgroup per CU, one wave per SIMD, and every wave walks
// "stage tile k -> barrier -> issue loads of tile k+1 -> MFMA sweep -> barrier".  With one wave per SIMD nothing hides the
// loader: the matrix pipe idles through staging, the barriers and the load issue (profiles/r03_pmc_sq.md: MFMA pipe busy
// 0.47-0.49 on the 16 / 32-channel classes, 0.66 on the 64-channel class; tools/gpu_wgrad_phases.py: 23 % of a tile
// outside the sweep).  Here the two kinds of work run in DIFFERENT waves of one persistent 8- or 12-wave workgroup per CU — the
// recipe conv_ws.hip proved on the same layers' forward / data-gradient launches:
//
//   waves 0..3 (consumers, one per SIMD): ds_read_b32 operand fetches + v_mfma_f32_16x16x4_f32 only — the sweep of
//                          wgrad_kernel.h, same (wm, wn, wk) wave grid, same row / k-step / tap order -> the partial rows
//                          are BIT-IDENTICAL to the one-kind-of-wave kernel's (tests compare them);
//   waves 4..7 / 4..11 (producers, s_setprio 3): global loads of tile k+2 into registers; BatchNorm affine (+ LeakyReLU of a
//                          ResBlock input) + zero padding of x, the fused BN / LeakyReLU backward of dy and the bias-gradient
//                          partial sums while staging tile k+1 into the OTHER LDS image (double-buffered).
//
// ONE __syncthreads() per tile: at barrier k the consumers have finished reading image k & 1 and the producers have
// finished writing image (k + 1) & 1.  LDS: 2 x (x halo tile + dpre tile) = 85 KB (16 + 16 -> 16 at TH = 8), 66 KB
// (32 -> 32, TH = 4), 109 KB (64 -> 64-column block, TH = 4).  Same arguments, plan (plan_wgrad), split-K rows and
// partial-row layout as wgrad_kernel.h; wgrad.hip routes a launch here when amx_wgrad_ws_supported() says so
// (AMX_WGRAD_WS=0 switches it off for A/B measurements).
#include "wgrad_kernel.h"

#ifndef AMX_WGRAD_WS_PRIO
#define AMX_WGRAD_WS_PRIO 3
#endif

// AMX_WGRAD_PROFILE (dev builds, tools/gpu_wgrad_ws_phases.py): [workgroup][12 wave slots][8] shader-clock totals — consumers:
// 0 sweep, 1 barrier wait; producers: 2 stage, 3 issue, 4 barrier wait; 6 tiles, 7 lifetime.
// AMX_WGRAD_WS_WG2: 1 = the 64-channel class (WM 4) is built for TWO workgroups' worth of registers per CU (<= 128 per wave
// with 8 waves), so that two 128-register convolution waves of the main stream still fit every SIMD next to it
// (experiment switch, tools/build_variant_lib.sh).
#ifndef AMX_WGRAD_WS_WG2
#define AMX_WGRAD_WS_WG2 0
#endif
template <int NT, int WM, int WN, int TH, int LAT, int NP>
__global__ __launch_bounds__(256 + NP, (AMX_WGRAD_WS_WG2 && WM == 4 && NP == 256) ? 4 : 1) void wgrad_ws_kernel(WgradArgs a) {
    constexpr int TAPS = 9;
    constexpr int LS = LAT ? LAT : 1;
    constexpr int CIB = 16 * WM;
    constexpr int CG = CIB / 4;
    constexpr int SX = (CIB % 32 == 16) ? CIB : CIB + 16;         // == 16 mod 32
    constexpr int IW = TW + 2, IH = TH + 2;
    constexpr int NPIX_X = IH * IW;
    constexpr int XLD = (NPIX_X * CG + NP - 1) / NP;              // NP = producer threads (4 or 8 waves)
    constexpr int COB = 16 * NT * WN;                             // the wave grid is compile-time here: WM x WN x WK = 4
    constexpr int WK = 4 / (WM * WN);
    constexpr int NSTEP = (TH / WK) * (TW / 4);                  // k-steps (4 pixels) of a consumer wave per tile
    constexpr int DG = COB / 4;
    constexpr int dg_shift = DG == 4 ? 2 : (DG == 8 ? 3 : 4);
    constexpr int SD = (COB % 32 == 16) ? COB : COB + 16;
    constexpr int nd4 = TH * TW * DG;
    constexpr int DLD_MAX = (nd4 + NP - 1) / NP;
    constexpr int XF = NPIX_X * SX;                               // floats of one x image
    constexpr int BUF = XF + TH * TW * SD;                        // floats of one (x, dpre) image pair
    AMX_DYN_SMEM(float, smem);

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef AMX_EMU
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int cb = blockIdx.y / a.co_blocks, ob = blockIdx.y % a.co_blocks;
    const int ci0 = cb * CIB, co0 = ob * COB;
    const int ntiles = a.tiles_x * a.tiles_y * a.N * LS * LS;
    float4 bsum = make_float4(0, 0, 0, 0);
#ifdef AMX_WGRAD_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pl = __builtin_amdgcn_s_memtime();
    const unsigned long long pstart = pl;
#endif

    if (wave < 4) {
        // ------------------------------------------------------------------ consumers: operand reads + MFMA only
        const int p = lane & 15, g = lane >> 4;
        const int wm = wave % WM;
        const int wn = (wave / WM) % WN;
        const int wk = wave / (WM * WN);
        f32x4 acc[TAPS][NT];
        #pragma unroll
        for (int t = 0; t < TAPS; ++t)
            #pragma unroll
            for (int q = 0; q < NT; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                          // the first tile is staged
        WG_TICK(1);
        int k = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += a.ksplit, ++k) {
            const float* s_x = smem + (size_t)(k & 1) * BUF;
            const float* s_d = s_x + XF;
            // Software-pipelined sweep: the 9 + NT operand values of k-step s + 1 are fetched into a second register
            // set while the 9 * NT MFMAs of step s issue (left to itself the compiler re-uses two operand registers and
            // waits for every ds_read right behind it — an LDS round trip exposed per 2-3 MFMAs, with ONE wave per SIMD).
            // Same (row, k-step, tap) order as wgrad_kernel.h -> bit-identical accumulators.
            float af[2][TAPS], bf[2][NT];
            // one lane base per image and tile; every operand address is that base + a compile-time offset (the top-left
            // halo pixel of the wave's first row, so that all offsets are >= 0: the ds_read immediate is unsigned)
            const float* xl = s_x + (size_t)(wk * IW + g) * SX + wm * 16 + p;
            const float* dl = s_d + (size_t)(wk * TW + g) * SD + wn * NT * 16 + p;
            auto fetch = [&](int st, float (&fa)[TAPS], float (&fb)[NT]) {
                const int rr = (st / (TW / 4)) * WK, kx = st % (TW / 4);     // row relative to the wave's first, k-step
                #pragma unroll
                for (int q = 0; q < NT; ++q)
                    fb[q] = dl[(rr * TW + kx * 4) * SD + q * 16];
                #pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    fa[t] = xl[((rr + 1 + dy) * IW + kx * 4 + 1 + dx) * SX];
                }
            };
            fetch(0, af[0], bf[0]);
            #pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) fetch(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
                #pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    #pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st & 1][t], bf[st & 1][q], acc[t][q], 0, 0, 0);
#ifndef AMX_EMU
                // issue order: one operand fetch of the NEXT step behind every NT MFMAs of this one
                #pragma unroll
                for (int t = 0; t < TAPS + NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);     // NT MFMA
                }
#endif
            }
            WG_TICK(0);
            __syncthreads();
            WG_TICK(1);
#ifdef AMX_WGRAD_PROFILE
            pt[6] += 1;
#endif
        }
        // D fragment: row (ci) = 4*g + reg, col (co) = p.  Partial row index = blockIdx.x * WK + wk.
        const size_t row = (size_t)blockIdx.x * WK + wk;
        #pragma unroll
        for (int t = 0; t < TAPS; ++t)
            #pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int co = co0 + (wn * NT + q) * 16 + p;
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + wm * 16 + 4 * g + r;
                    if (ci < a.ci_pad && co < a.co_pad)
                        a.part[((row * TAPS + t) * a.ci_pad + ci) * a.co_pad + co] = acc[t][q][r];
                }
            }
    } else {
        // ------------------------------------------------------------------ producers: loads, transforms, staging
#if !defined(AMX_EMU) && AMX_WGRAD_WS_PRIO
        __builtin_amdgcn_s_setprio(AMX_WGRAD_WS_PRIO);            // (conv_ws.hip: the arbiter serves the oldest ready wave,
#endif                                                            //  and a consumer always has an MFMA waiting for the pipe)
        const int ptid = tid - 256;
        const int xg = ptid % CG;
        const int ch = ci0 + xg * 4;
        const float* xsrc = nullptr; int xCs = 0, xc = 0;
        float4 r_sc = make_float4(1, 1, 1, 1), r_sh = make_float4(0, 0, 0, 0);
        float r_islope = 1.f;
        if (ch < a.C0s) { xsrc = a.x0; xCs = a.C0s; xc = ch; r_islope = a.in_slope0; if (a.sc0) { r_sc = amx_ld4(a.sc0 + xc); r_sh = amx_ld4(a.sh0 + xc); } }
        else if (ch - a.C0s < a.C1s) { xsrc = a.x1; xCs = a.C1s; xc = ch - a.C0s; r_islope = a.in_slope1; if (a.sc1) { r_sc = amx_ld4(a.sc1 + xc); r_sh = amx_ld4(a.sh1 + xc); } }

        float4 xr[XLD];
        float4 dr[DLD_MAX];
        unsigned xvalid = 0;
        long d_off[DLD_MAX];
        // tile-independent load descriptors (wgrad_kernel.h): element offset of a slot relative to the tile origin and
        // its (iy << 8) | ix image-space offsets, or -1
        int x_rel[XLD], x_yx[XLD];
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int pix = (ptid + i * NP) / CG;
            x_yx[i] = -1; x_rel[i] = 0;
            if (pix < NPIX_X && xsrc) {
                const int iy = pix / IW, ix = pix - iy * IW;
                x_yx[i] = ((iy * LS) << 8) | (ix * LS);
                x_rel[i] = (iy * LS * a.W + ix * LS) * xCs + xc;
            }
        }
        int d_rel[DLD_MAX], d_yx[DLD_MAX];
        #pragma unroll
        for (int i = 0; i < DLD_MAX; ++i) {
            const int idx = ptid + i * NP;
            d_yx[i] = -1; d_rel[i] = 0;
            if (idx < nd4) {
                const int pix = idx >> dg_shift, dg = idx & (DG - 1);
                const int iy = pix / TW, ix = pix - iy * TW;
                const int c = co0 + dg * 4;
                if (c < a.Dos) { d_yx[i] = ((iy * LS) << 8) | (ix * LS); d_rel[i] = (iy * LS * a.W + ix * LS) * a.Dos + c; }
            }
        }
        // every load is UNCONDITIONAL (clamped to a safe address, zeroed when staged): see wgrad_kernel.h / conv_ws.hip
        const float* xsafe = xsrc ? xsrc + xc : a.x0;
        auto issue = [&](int n, int ty, int tx, int rr) {
            const int ry = LAT ? rr / LS : 0, rx = LAT ? rr - ry * LS : 0;
            const int gy0 = ry + (ty * TH - 1) * LS, gx0 = rx + (tx * TW - 1) * LS;
            const long xbase = ((long)(n * a.H + gy0) * a.W + gx0) * xCs;
            xvalid = 0;
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int gy = gy0 + (x_yx[i] >> 8), gx = gx0 + (x_yx[i] & 255);
                const bool ok = x_yx[i] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const float* src = ok ? xsrc + (xbase + x_rel[i]) : xsafe;
                xr[i] = amx_ld4(src);
                xvalid |= (ok ? 1u : 0u) << i;
            }
            const int dy0 = ry + ty * TH * LS, dx0 = rx + tx * TW * LS;
            const long dbase = ((long)(n * a.H + dy0) * a.W + dx0) * a.Dos;
            #pragma unroll
            for (int i = 0; i < DLD_MAX; ++i) {
                const int gy = dy0 + (d_yx[i] >> 8), gx = dx0 + (d_yx[i] & 255);
                const bool ok = d_yx[i] >= 0 && gy < a.H && gx < a.W;
                const long o = ok ? dbase + d_rel[i] : 0;
                dr[i] = amx_ld4(a.dpre + o);
                d_off[i] = ok ? o : -1;
            }
        };
        auto stage = [&](int buf) {
            float* s_x = smem + (size_t)buf * BUF;
            float* s_d = s_x + XF;
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int pix = (ptid + i * NP) / CG;
                if (pix < NPIX_X) {
                    float4 v = xr[i];
                    v.x = fmaf(v.x, r_sc.x, r_sh.x); v.y = fmaf(v.y, r_sc.y, r_sh.y);
                    v.z = fmaf(v.z, r_sc.z, r_sh.z); v.w = fmaf(v.w, r_sc.w, r_sh.w);
                    if (r_islope != 1.f) {
                        v.x = v.x > 0.f ? v.x : v.x * r_islope; v.y = v.y > 0.f ? v.y : v.y * r_islope;
                        v.z = v.z > 0.f ? v.z : v.z * r_islope; v.w = v.w > 0.f ? v.w : v.w * r_islope;
                    }
                    if (!(xvalid & (1u << i))) v = make_float4(0.f, 0.f, 0.f, 0.f);     // zero padding (AFTER the affine)
                    amx_st4(s_x + (size_t)pix * SX + xg * 4, v);
                }
            }
            #pragma unroll
            for (int i = 0; i < DLD_MAX; ++i) {
                const int idx = ptid + i * NP;
                if (idx < nd4) {
                    const int pix = idx >> dg_shift, dg = idx & (DG - 1);
                    float4 v = dr[i];
                    if (d_off[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.aux) {
                        if (d_off[i] >= 0) {                      // dpre = lrelu'(a) * (k1*dy + k2*a + k3)
                            const int c = co0 + dg * 4;
                            float4 c1 = make_float4(1, 1, 1, 1), c2 = make_float4(0, 0, 0, 0), c3 = c2;
                            if (a.k1) { c1 = amx_ld4(a.k1 + c); c2 = amx_ld4(a.k2 + c); c3 = amx_ld4(a.k3 + c); }
                            const float4 t = amx_ld4(a.aux + d_off[i]);
                            v.x = (t.x > 0.f ? 1.f : a.bslope) * fmaf(c1.x, v.x, fmaf(c2.x, t.x, c3.x));
                            v.y = (t.y > 0.f ? 1.f : a.bslope) * fmaf(c1.y, v.y, fmaf(c2.y, t.y, c3.y));
                            v.z = (t.z > 0.f ? 1.f : a.bslope) * fmaf(c1.z, v.z, fmaf(c2.z, t.z, c3.z));
                            v.w = (t.w > 0.f ? 1.f : a.bslope) * fmaf(c1.w, v.w, fmaf(c2.w, t.w, c3.w));
                        } else {
                            v = make_float4(0, 0, 0, 0);
                        }
                    }
                    if (a.bpart && cb == 0) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
                    amx_st4(s_d + (size_t)pix * SD + dg * 4, v);
                }
            }
        };
        // (residue, tx, ty, n) of the tile whose loads are issued next, advanced by the split-K stride with carries
        constexpr int RR = LS * LS;
        const int tpi = a.tiles_x * a.tiles_y;
        int tile = blockIdx.x;
        int trr = tile % RR, ttx = (tile / RR) % a.tiles_x, tty = (tile / RR / a.tiles_x) % a.tiles_y, ttn = tile / RR / tpi;
        const int srr = a.ksplit % RR, stx = (a.ksplit / RR) % a.tiles_x, sty = (a.ksplit / RR / a.tiles_x) % a.tiles_y,
                  stn = a.ksplit / RR / tpi;
        auto advance = [&]() {
            trr += srr; int carry = trr >= RR ? 1 : 0; trr -= carry ? RR : 0;
            ttx += stx + carry; carry = ttx >= a.tiles_x ? 1 : 0; ttx -= carry ? a.tiles_x : 0;
            tty += sty + carry; carry = tty >= a.tiles_y ? 1 : 0; tty -= carry ? a.tiles_y : 0;
            ttn += stn + carry;
        };
        if (tile < ntiles) {
            issue(ttn, tty, ttx, trr);
            stage(0);
            if (tile + a.ksplit < ntiles) { advance(); issue(ttn, tty, ttx, trr); }
        }
        WG_TICK(2);
        __syncthreads();
        WG_TICK(4);
        int k = 0;
        for (; tile < ntiles; tile += a.ksplit, ++k) {
            if (tile + a.ksplit < ntiles) {
                stage((k + 1) & 1);                               // tile k+1: its image was last read for tile k-1
                WG_TICK(2);
                if (tile + 2 * a.ksplit < ntiles) { advance(); issue(ttn, tty, ttx, trr); }
                WG_TICK(3);
            }
            __syncthreads();
            WG_TICK(4);
        }
    }

    if (a.bpart && cb == 0) {
        // producer threads with equal ptid % DG staged the same channel group: fixed-order sum over them
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem);
        if (tid >= 256) red[tid - 256] = bsum;
        __syncthreads();
        if (tid < DG) {
            float4 t = make_float4(0, 0, 0, 0);
            for (int q = tid; q < NP; q += DG) { const float4 u = red[q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const int c = co0 + tid * 4;
            const float tv[4] = {t.x, t.y, t.z, t.w};
            for (int e = 0; e < 4; ++e) if (c + e < a.co_pad) a.bpart[(size_t)blockIdx.x * a.co_pad + c + e] = tv[e];
        }
    }
#ifdef AMX_WGRAD_PROFILE
    pt[7] = __builtin_amdgcn_s_memtime() - pstart;
    if (a.prof && lane == 0)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 12 + wave) * 8 + i] = pt[i];
#endif
}

template <int NT, int WM, int WN, int TH, int LAT, int NP>
static int launch_wgrad_ws_np(const WgradArgs& a, hipStream_t stream) {
    constexpr int CIB = 16 * WM;
    constexpr int SX = (CIB % 32 == 16) ? CIB : CIB + 16;
    constexpr int COB = 16 * NT * WN;
    constexpr int SD = (COB % 32 == 16) ? COB : COB + 16;
    if (a.WN != WN || a.WK != 4 / (WM * WN)) AMX_BADARG(9);
    const size_t lds = 2 * ((size_t)(TH + 2) * (TW + 2) * SX + (size_t)TH * TW * SD) * sizeof(float);
    dim3 grid(a.ksplit, amx_ceil_div(a.ci_pad, CIB) * a.co_blocks);
    AMX_ALLOW_160K_LDS(wgrad_ws_kernel<NT, WM, WN, TH, LAT, NP>);
    AMX_LAUNCH((wgrad_ws_kernel<NT, WM, WN, TH, LAT, NP>), grid, dim3(256 + NP), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Producer waves: 8 on the <= 32-input-channel classes (with 4 the producers' stage + issue phases fill the whole tile time
// there and the consumers wait 13-24 % of their life at the barrier; with 8 they wait 2-9 %: 0.66 -> 0.68, 0.60 -> 0.66,
// 0.74 -> 0.77 of peak stand-alone), 4 on the 64-channel class (no difference: 0.81-0.83 either way);
// profiles/r04_logs/r04_wgrad_ws_phases2.log, r04_wgrad_ws_np_ab.log (measured with a since-removed override).
template <int NT, int WM, int WN, int TH, int LAT>
static int launch_wgrad_ws(const WgradArgs& a, hipStream_t stream) {
    constexpr int np = WM < 4 ? 8 : 4;
    return np == 4 ? launch_wgrad_ws_np<NT, WM, WN, TH, LAT, 256>(a, stream) : launch_wgrad_ws_np<NT, WM, WN, TH, LAT, 512>(a, stream);
}

// Which launches the wave-specialised kernel takes: the plain 3x3 classes of plan_wgrad (lattice-mode dilated layers and
// the 1x1 / halo classes stay on wgrad_kernel.h).  AMX_WGRAD_WS: 0 = off, 1 (default) = the classes of AMX_WGRAD_WS_MASK,
// a bit mask over the wave layouts: 1 = WM 1 (16 input channels), 2 = WM 2 (32), 4 = WM 4 (>= 64).
// Default mask 3: stand-alone the kernel wins on every class (+20 % on the 64-channel one), but inside the training step a
// persistent 8-wave / 109 KB workgroup of that class keeps the main stream's convolution workgroups off its CU while the
// one-wave-per-SIMD kernel of wgrad_kernel.h lets them in: U-Net step 17.40 ms with mask 7, 17.25 with mask 3, 18.04 with
// mask 4 (profiles/r04_logs/r04_step_ab5.log; the thin classes are where the loaders' fused BatchNorm backward needs the
// producer waves: 18.24 ms without the wave-specialised kernel at all).
int amx_wgrad_ws_mask() {
    const AmxKnobs& kn = amx_knobs();
    return kn.wgrad_ws <= 0 ? 0 : (kn.wgrad_ws_mask & 7);
}

bool amx_wgrad_ws_supported(const WgradArgs& a, int taps, int dil, int lat, int nt, int wm, int th) {
    if (taps != 9 || dil != 1 || lat) return false;
    if (!(amx_wgrad_ws_mask() & wm)) {
        // The 64-channel class (mask bit 4, off by default) still takes layers of at least 128 image rows: measured per
        // size inside the step (profiles/r04_logs/r04_step_ab8.log) the 256^2 and 128^2 launches cost nothing there
        // (17.21-17.25 against 17.24 ms) and are 20 % faster on their own, the 64^2 bottleneck layers lose 0.05-0.07 ms.
        // AMX_WGRAD_WS_WM4 = h > 0: at least h rows (default 128); h < 0: at most -h rows; 0: never.
        const int h = (wm == 4 && amx_wgrad_ws_mask()) ? amx_knobs().wgrad_ws_wm4 : 0;
        if (!(h > 0 && a.H >= h) && !(h < 0 && a.H <= -h)) return false;
    }
    if (nt == 1) return th == 8;
    return th == 4 || (th == 8 && wm == 1 && a.WN == 1);
}

static std::atomic<long> wgrad_ws_launches{0};
extern "C" long amx_conv2d_wgrad_ws_launches(void) { return wgrad_ws_launches.load(std::memory_order_relaxed); }

int amx_wgrad_launch_ws(const WgradArgs& a, int nt, int wm, int th, hipStream_t s) {
    wgrad_ws_launches.fetch_add(1, std::memory_order_relaxed);
    if (nt == 1) {                                                // 16 output channels: one cout tile, WN = 1
        if (wm == 1) return launch_wgrad_ws<1, 1, 1, 8, 0>(a, s);
        if (wm == 2) return launch_wgrad_ws<1, 2, 1, 8, 0>(a, s);
        return launch_wgrad_ws<1, 4, 1, 8, 0>(a, s);
    }
    if (wm == 1 && th == 8) return launch_wgrad_ws<2, 1, 1, 8, 0>(a, s);
    if (wm == 1) return a.WN == 2 ? launch_wgrad_ws<2, 1, 2, 4, 0>(a, s) : launch_wgrad_ws<2, 1, 1, 4, 0>(a, s);
    if (wm == 2) return a.WN == 2 ? launch_wgrad_ws<2, 2, 2, 4, 0>(a, s) : launch_wgrad_ws<2, 2, 1, 4, 0>(a, s);
    return launch_wgrad_ws<2, 4, 1, 4, 0>(a, s);
}
