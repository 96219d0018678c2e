// wgrad_kernel.h — weight gradient of the 3x3 / dilated / 1x1 convolutions on fp32 MFMA (kernel + launcher;
// instantiated by wgrad.hip and, for the lattice mode of dilations 2 / 4 / 6, wgrad_lat{2,4,6}.hip).
//
//   dW[co][ci][tap] = sum over pixels p of  dpre[p][co] * xin[p + tap*dil][ci]
// where xin is the layer input as the forward pass saw it: BN affine of the producer applied on load,
// two concatenated sources (skip | upsampled), zero padding.   (autograd of nn.Conv2d in
// atomai/nets/blocks.py:63-67, 304-310; the reference gets it from ATen's conv backward.)
//
// GEMM view per tap: M = ci (16-tiles), N = co (16-tiles), K = pixels.  v_mfma_f32_16x16x4_f32 with
// A[i = ci][k = pixel], B[k = pixel][j = co]: one k-step = 4 consecutive pixels of an image row.
// A workgroup keeps an (8 x 16)-pixel tile of dpre and the matching halo tile of xin in LDS
// (pixel-major, channel stride == 16 mod 32 floats so that the fragment reads `ds_read_b32` are
// conflict free) and sweeps all taps against the SAME B fragments, so 9*NT MFMAs are issued per
// (NT + 9) LDS reads.  Waves split the ci tiles (WM), co tiles (WN) and pixel rows (WK).
// Split-K over workgroups; every (workgroup, wk) writes its own partial row which
// amx_wgrad_reduce (conv1.hip) sums in fp64 -> deterministic, no float atomics.
#pragma once
#include "amx_device.h"

#include <cstdlib>
#define TW 16

struct WgradArgs {
    const float* x0; const float* sc0; const float* sh0; int C0s;
    const float* x1; const float* sc1; const float* sh1; int C1s;
    float in_slope0, in_slope1;      // LeakyReLU after the on-load affine of source 0 / 1 (1.0f == none)
    const float* dpre; int Dos;      // stored channels of dpre (or of dy when aux != nullptr)
    const float* aux;                // activation a: dpre = lrelu'(a) * (k1*dy + k2*a + k3) formed while loading
    const float* k1; const float* k2; const float* k3; float bslope;
    float* bpart;                    // [ksplit][co_pad] per-workgroup sums of dpre (bias gradient) or nullptr
    float* part;                     // [rows][taps][ci_pad][co_pad]
    unsigned long long* prof;        // dev builds: per-wave phase clocks (or nullptr)
    int N, H, W, dil;
    int ci_pad, co_pad;              // multiples of 16
    int WN, WK;                      // wave grid (WM is a template parameter)
    int ksplit, tiles_x, tiles_y;
    int co_blocks;
};

// AMX_WGRAD_PROFILE (dev builds only, tools/gpu_wgrad_phases.py): every wave accumulates the shader clocks it spends in
// each phase of its tile loop and writes [workgroup][wave][8] 64-bit totals (stage, barrier, issue, mfma, barrier, tail,
// tiles, lifetime) to the buffer set through amx_wgrad_set_profile_buffer.
#ifdef AMX_WGRAD_PROFILE
extern void* amx_wgrad_profile_buffer;
#define WG_TICK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt[i] += n_ - pl; pl = n_; } while (0)
#else
#define WG_TICK(i) do { } while (0)
#endif
#ifndef AMX_WGRAD_EXACT
#define AMX_WGRAD_EXACT 1        // compile-time experiment switch (tools/build_variant_lib.sh): 0 = runtime halo as in round 1
#endif
template <int TAPS, int NT, int WM, int MAXHALO, int TH, int LAT = 0>
// Forcing two waves per SIMD for the wide variant (191 + 72 registers -> 256 with 6 spills) was measured in-step with
// tools/gpu_lib_ab.py: 20.82 ms (256 workgroups) / 20.50 ms (384) against 20.34 ms for one wave per SIMD -> rejected.
// Re-measured in round 3 with every 4-row class bounded to 256 registers (so that two 128-register convolution waves of
// the main stream could share a SIMD with a weight-gradient wave instead of one): 18.51 -> 19.03 ms per step
// (profiles/r03_wgrad_regs_ab.log) — the convolution waves then take MFMA issue slots from the weight-gradient stream,
// whose kernels already span the whole backward pass (11.7 ms in-step for 5.9 ms stand-alone).
#ifndef AMX_WGRAD_WAVES
#define AMX_WGRAD_WAVES 1
#endif
__global__ __launch_bounds__(256, (TAPS == 9 && NT == 2 && WM == 4 && TH == 4) ? AMX_WGRAD_WAVES : 1) void wgrad_kernel(WgradArgs a) {
    static_assert(LAT == 0 || (TAPS == 9 && MAXHALO == 1), "lattice mode runs the plain 3x3 geometry");
    // LAT (dilation 2 / 4 / 6): the dilated convolution is LAT*LAT plain 3x3 convolutions on the residue-class
    // sub-images x[ry::LAT, rx::LAT] (conv_kernel.h); a tile is TH x 16 pixels of ONE sub-lattice, so the LDS images and
    // the MFMA sweep are those of the plain kernel and only the global addresses of the loaders are strided.
    constexpr int LS = LAT ? LAT : 1;
    constexpr int CIB = 16 * WM;
    constexpr int CG = CIB / 4;                                   // float4 groups per pixel (x)
    constexpr int SX = (CIB % 32 == 16) ? CIB : CIB + 16;         // == 16 mod 32
    constexpr int MAXPIX = (TH + 2 * MAXHALO) * (TW + 2 * MAXHALO);
    constexpr int XLD = (MAXPIX * CG + 255) / 256;
    AMX_DYN_SMEM(float, smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    // plain 3x3 (MAXHALO == 1 is dispatched for dilation 1 only): compile-time tile geometry, so the per-tile index
    // math of the loaders divides by constants (a runtime integer division is ~25 VALU instructions, and with one
    // wave per SIMD nothing hides the staging phase)
    const int halo = (TAPS == 9) ? (((AMX_WGRAD_EXACT && MAXHALO == 1) || LAT) ? 1 : a.dil) : 0;
    const int IW = TW + 2 * halo, IH = TH + 2 * halo;
    const int COB = 16 * NT * a.WN;
    const int DG = COB / 4;                                       // float4 groups per pixel (dpre): 4, 8 or 16
    const int dg_shift = DG == 4 ? 2 : (DG == 8 ? 3 : 4);
    const int SD = (COB % 32 == 16) ? COB : COB + 16;
    float* s_x = smem;                                            // [IH*IW][SX]
    float* s_d = smem + (size_t)IH * IW * SX;                     // [TH*TW][SD]

    const int wm = wave % WM;
    const int wn = (wave / WM) % a.WN;
    const int wk = wave / (WM * a.WN);
    const int cb = blockIdx.y / a.co_blocks, ob = blockIdx.y % a.co_blocks;
    const int ci0 = cb * CIB;                                     // first concat-padded input channel
    const int co0 = ob * COB;

    // x loader: this thread always handles channel group xg of the block
    const int xg = tid % CG;
    const int ch = ci0 + xg * 4;
    const float* xsrc = nullptr; int xCs = 0, xc = 0;
    float4 r_sc = make_float4(1, 1, 1, 1), r_sh = make_float4(0, 0, 0, 0);
    float r_islope = 1.f;
    if (ch < a.C0s) { xsrc = a.x0; xCs = a.C0s; xc = ch; r_islope = a.in_slope0; if (a.sc0) { r_sc = amx_ld4(a.sc0 + xc); r_sh = amx_ld4(a.sh0 + xc); } }
    else if (ch - a.C0s < a.C1s) { xsrc = a.x1; xCs = a.C1s; xc = ch - a.C0s; r_islope = a.in_slope1; if (a.sc1) { r_sc = amx_ld4(a.sc1 + xc); r_sh = amx_ld4(a.sh1 + xc); } }
    const int npix_x = IH * IW;
    const int nd4 = TH * TW * DG;                                 // float4 loads of the dpre tile
    constexpr int DLD_MAX = (TH * TW * 16 + 255) / 256;           // COB <= 64 -> DG <= 16

    float4 xr[XLD];
    float4 dr[DLD_MAX];
    unsigned xvalid = 0;
    float4 bsum = make_float4(0, 0, 0, 0);   // bias-gradient partial of this thread's channel group (ci-block 0)
    long d_off[DLD_MAX];                     // element offsets of the dy values held in dr (or -1)

    // Tile-independent load descriptors.  One workgroup walks ntiles / ksplit tiles with ONE wave per SIMD, so every
    // instruction of the loaders is exposed (tools/gpu_wgrad_phases.py: issuing 9 loads cost 2400 clocks per tile when
    // the slot -> pixel index math was redone per tile).  Per tile remain: two bases, and per load an add and the
    // bounds compares.
    int x_rel[XLD], x_yx[XLD];               // element offset of the slot relative to the tile origin; (iy << 8) | ix or -1
    #pragma unroll
    for (int i = 0; i < XLD; ++i) {
        const int pix = (tid + i * 256) / CG;
        x_yx[i] = -1; x_rel[i] = 0;
        if (pix < npix_x && xsrc) {
            const int iy = pix / IW, ix = pix - iy * IW;
            x_yx[i] = ((iy * LS) << 8) | (ix * LS);            // image-space offsets from the tile origin
            x_rel[i] = (iy * LS * a.W + ix * LS) * xCs + xc;
        }
    }
    int d_rel[DLD_MAX], d_yx[DLD_MAX];
    #pragma unroll
    for (int i = 0; i < DLD_MAX; ++i) {
        const int idx = tid + i * 256;
        d_yx[i] = -1; d_rel[i] = 0;
        if (idx < nd4) {
            const int pix = idx >> dg_shift, dg = idx & (DG - 1);
            const int iy = pix / TW, ix = pix - iy * TW;
            const int c = co0 + dg * 4;
            if (c < a.Dos) { d_yx[i] = ((iy * LS) << 8) | (ix * LS); d_rel[i] = (iy * LS * a.W + ix * LS) * a.Dos + c; }
        }
    }

    // Every load is UNCONDITIONAL (conv_ws.hip found the same): a load under a bounds predicate whose destination was
    // zeroed first makes the compiler merge old and new register values right behind the load — the wave then waits for
    // the data where the load was issued, and with ONE wave per SIMD that latency is part of the "issue" phase (1728 -> 1541
    // clocks per tile on the >= 32-channel classes, profiles/r03_wgrad_phases.log -> r03_wgrad_phases_uncond.log; U-Net
    // step 18.19 -> 18.11 ms).  Slots outside the image (or without a
    // source) read a safe address instead and are zeroed when the tile is staged.
    const float* xsafe = xsrc ? xsrc + xc : a.x0;
    auto issue = [&](int n, int ty, int tx, int rr) {
        const int ry = LAT ? rr / LS : 0, rx = LAT ? rr - ry * LS : 0;     // residue class of this tile (lattice mode)
        const int gy0 = ry + (ty * TH - halo) * LS, gx0 = rx + (tx * TW - halo) * LS;   // image-space origin incl. halo
        const long xbase = ((long)(n * a.H + gy0) * a.W + gx0) * xCs;      // (may be negative: halo rows of image 0)
        // (A wave-uniform fast path for tiles whose halo lies inside the image — no per-slot bounds arithmetic — was
        // measured: the two code paths writing the same registers bring the wait-at-the-merge back, issue phase 1541 ->
        // 2397 clocks per tile, step 18.11 -> 18.48 ms; profiles/r03_wgrad_phases_interior.log.)
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
    auto stage = [&]() {
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int pix = (tid + i * 256) / CG;
            if (pix < npix_x) {
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
            const int idx = tid + i * 256;
            if (idx < nd4) {
                const int pix = idx >> dg_shift, dg = idx & (DG - 1);
                float4 v = dr[i];
                if (d_off[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.aux) {
                    // dpre = lrelu'(a) * (k1*dy + k2*a + k3); a is fetched here rather than prefetched so that the
                    // register footprint (hence the number of co-resident workgroups) stays that of the plain kernel
                    if (d_off[i] >= 0) {
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

    f32x4 acc[TAPS][NT];
    #pragma unroll
    for (int t = 0; t < TAPS; ++t)
        #pragma unroll
        for (int q = 0; q < NT; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = a.tiles_x * a.tiles_y * a.N * LS * LS;
    int tile = blockIdx.x;
    // (residue, tx, ty, n) of this workgroup's current tile, advanced by the split-K stride with carries (no division
    // per tile); the residue classes of one tile position are consecutive tiles
    constexpr int RR = LS * LS;
    const int tpi = a.tiles_x * a.tiles_y;
    int trr = tile % RR, ttx = (tile / RR) % a.tiles_x, tty = (tile / RR / a.tiles_x) % a.tiles_y, ttn = tile / RR / tpi;
    const int srr = a.ksplit % RR, stx = (a.ksplit / RR) % a.tiles_x, sty = (a.ksplit / RR / a.tiles_x) % a.tiles_y,
              stn = a.ksplit / RR / tpi;
#ifdef AMX_WGRAD_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pl = __builtin_amdgcn_s_memtime();
    const unsigned long long pstart = pl;
#endif
    if (tile < ntiles) issue(ttn, tty, ttx, trr);
    for (; tile < ntiles; tile += a.ksplit) {
        WG_TICK(2);
        stage();
        WG_TICK(0);
        __syncthreads();
        WG_TICK(1);
        if (tile + a.ksplit < ntiles) {
            trr += srr; int carry = trr >= RR ? 1 : 0; trr -= carry ? RR : 0;
            ttx += stx + carry; carry = ttx >= a.tiles_x ? 1 : 0; ttx -= carry ? a.tiles_x : 0;
            tty += sty + carry; carry = tty >= a.tiles_y ? 1 : 0; tty -= carry ? a.tiles_y : 0;
            ttn += stn + carry;
            issue(ttn, tty, ttx, trr);
        }
        WG_TICK(2);
        // (Rejected, round 2: rows of this sweep unrolled 1 / 2 / 4-fold with a compile-time trip count — stand-alone
        // +3..5 % on the >= 64-channel classes, but 30-60 more registers per wave, and inside the training step, where
        // these waves share the SIMDs with the data-gradient kernels, 19.35 -> 19.95 / 20.05 / 20.55 ms.)
        for (int r = wk; r < TH; r += a.WK) {
            #pragma unroll
            for (int kx = 0; kx < TW / 4; ++kx) {
                float bf[NT];
                #pragma unroll
                for (int q = 0; q < NT; ++q)
                    bf[q] = s_d[(size_t)(r * TW + kx * 4 + g) * SD + (wn * NT + q) * 16 + p];
                #pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int dy = (TAPS == 9) ? (t / 3 - 1) * (LAT ? 1 : a.dil) : 0;
                    const int dx = (TAPS == 9) ? (t % 3 - 1) * (LAT ? 1 : a.dil) : 0;
                    const float af = s_x[(size_t)((r + halo + dy) * IW + kx * 4 + g + halo + dx) * SX + wm * 16 + p];
                    #pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[q], acc[t][q], 0, 0, 0);
                }
            }
        }
        WG_TICK(3);
        __syncthreads();
        WG_TICK(4);
#ifdef AMX_WGRAD_PROFILE
        pt[6] += 1;
#endif
    }

    if (a.bpart && cb == 0) {
        // threads with equal tid % DG staged the same channel group: fixed-order sum over them
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem);
        red[tid] = bsum;
        __syncthreads();
        if (tid < DG) {
            float4 t = make_float4(0, 0, 0, 0);
            for (int q = tid; q < 256; q += DG) { const float4 u = red[q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const int c = co0 + tid * 4;
            const float tv[4] = {t.x, t.y, t.z, t.w};
            for (int e = 0; e < 4; ++e) if (c + e < a.co_pad) a.bpart[(size_t)blockIdx.x * a.co_pad + c + e] = tv[e];
        }
    }
    // D fragment: row (ci) = 4*g + reg, col (co) = p.  Partial row index = blockIdx.x * WK + wk.
    const size_t row = (size_t)blockIdx.x * a.WK + wk;
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
#ifdef AMX_WGRAD_PROFILE
    WG_TICK(5);
    pt[7] = pl - pstart;
    if (a.prof && lane == 0)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + i] = pt[i];
#endif
}

template <int TAPS, int NT, int WM, int MAXHALO, int TH, int LAT = 0>
static int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    constexpr int CIB = 16 * WM;
    constexpr int SX = (CIB % 32 == 16) ? CIB : CIB + 16;
    const int halo = (TAPS == 9) ? (LAT ? 1 : a.dil) : 0;
    const int COB = 16 * NT * a.WN;
    const int SD = (COB % 32 == 16) ? COB : COB + 16;
    const size_t lds = ((size_t)(TH + 2 * halo) * (TW + 2 * halo) * SX + (size_t)TH * TW * SD) * sizeof(float);
    dim3 grid(a.ksplit, amx_ceil_div(a.ci_pad, CIB) * a.co_blocks);
    AMX_ALLOW_160K_LDS(wgrad_kernel<TAPS, NT, WM, MAXHALO, TH, LAT>);
    AMX_LAUNCH((wgrad_kernel<TAPS, NT, WM, MAXHALO, TH, LAT>), grid, dim3(256), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}


// wave-specialised (producer / consumer) kernel for the plain 3x3 classes: wgrad_ws.hip
int amx_wgrad_ws_mask();            // 0 = off; else the AMX_WGRAD_WS_MASK bits (1 = WM 1, 2 = WM 2, 4 = WM 4)
bool amx_wgrad_ws_supported(const WgradArgs& a, int taps, int dil, int lat, int nt, int wm, int th);
int amx_wgrad_launch_ws(const WgradArgs& a, int nt, int wm, int th, hipStream_t s);

// lattice-mode instantiation units (one per dilation: compiled in parallel); nt in {1, 2} with th = 8 / 4 as planned
int amx_wgrad_launch_lat2(const WgradArgs& a, int nt, int wm, hipStream_t s);
int amx_wgrad_launch_lat4(const WgradArgs& a, int nt, int wm, hipStream_t s);
int amx_wgrad_launch_lat6(const WgradArgs& a, int nt, int wm, hipStream_t s);
#define AMX_WGRAD_LAT_UNIT(D_)                                                        \
    int amx_wgrad_launch_lat##D_(const WgradArgs& a, int nt, int wm, hipStream_t s) {  \
        if (nt == 1) {                                                                \
            if (wm == 1) return launch_wgrad<9, 1, 1, 1, 8, D_>(a, s);                \
            if (wm == 2) return launch_wgrad<9, 1, 2, 1, 8, D_>(a, s);                \
            return launch_wgrad<9, 1, 4, 1, 8, D_>(a, s);                             \
        }                                                                             \
        if (wm == 1) return launch_wgrad<9, 2, 1, 1, 4, D_>(a, s);                    \
        if (wm == 2) return launch_wgrad<9, 2, 2, 1, 4, D_>(a, s);                    \
        return launch_wgrad<9, 2, 4, 1, 4, D_>(a, s);                                 \
    }
