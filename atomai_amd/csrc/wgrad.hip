// wgrad.hip — weight gradient of the 3x3 / dilated / 1x1 convolutions on fp32 MFMA.
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
static void* amx_wgrad_profile_buffer = nullptr;
extern "C" int amx_wgrad_set_profile_buffer(void* buf) { amx_wgrad_profile_buffer = buf; return 0; }
#define WG_TICK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt[i] += n_ - pl; pl = n_; } while (0)
#else
#define WG_TICK(i) do { } while (0)
#endif
#ifndef AMX_WGRAD_EXACT
#define AMX_WGRAD_EXACT 1        // compile-time experiment switch (tools/build_variant_lib.sh): 0 = runtime halo as in round 1
#endif
template <int TAPS, int NT, int WM, int MAXHALO, int TH>
// Forcing two waves per SIMD for the wide variant (191 + 72 registers -> 256 with 6 spills) was measured in-step with
// tools/gpu_lib_ab.py: 20.82 ms (256 workgroups) / 20.50 ms (384) against 20.34 ms for one wave per SIMD -> rejected.
#ifndef AMX_WGRAD_WAVES
#define AMX_WGRAD_WAVES 1
#endif
__global__ __launch_bounds__(256, (TAPS == 9 && NT == 2 && WM == 4 && TH == 4) ? AMX_WGRAD_WAVES : 1) void wgrad_kernel(WgradArgs a) {
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
    const int halo = (TAPS == 9) ? ((AMX_WGRAD_EXACT && MAXHALO == 1) ? 1 : a.dil) : 0;
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
            x_yx[i] = (iy << 8) | ix;
            x_rel[i] = (iy * a.W + ix) * xCs + xc;
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
            if (c < a.Dos) { d_yx[i] = (iy << 8) | ix; d_rel[i] = (iy * a.W + ix) * a.Dos + c; }
        }
    }

    auto issue = [&](int n, int ty, int tx) {
        const int gy0 = ty * TH - halo, gx0 = tx * TW - halo;
        const long xbase = ((long)(n * a.H + gy0) * a.W + gx0) * xCs;      // (may be negative: halo rows of image 0)
        xvalid = 0;
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            xr[i] = make_float4(0, 0, 0, 0);
            if (x_yx[i] >= 0) {
                const int gy = gy0 + (x_yx[i] >> 8), gx = gx0 + (x_yx[i] & 255);
                if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                    xr[i] = amx_ld4(xsrc + (xbase + x_rel[i]));
                    xvalid |= 1u << i;
                }
            }
        }
        const long dbase = ((long)(n * a.H + ty * TH) * a.W + tx * TW) * a.Dos;
        #pragma unroll
        for (int i = 0; i < DLD_MAX; ++i) {
            dr[i] = make_float4(0, 0, 0, 0);
            d_off[i] = -1;
            if (d_yx[i] >= 0) {
                const int gy = ty * TH + (d_yx[i] >> 8), gx = tx * TW + (d_yx[i] & 255);
                if (gy < a.H && gx < a.W) {
                    const long o = dbase + d_rel[i];
                    dr[i] = amx_ld4(a.dpre + o);
                    d_off[i] = o;
                }
            }
        }
    };
    auto stage = [&]() {
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int pix = (tid + i * 256) / CG;
            if (pix < npix_x) {
                float4 v = xr[i];
                if (xvalid & (1u << i)) {
                    v.x = fmaf(v.x, r_sc.x, r_sh.x); v.y = fmaf(v.y, r_sc.y, r_sh.y);
                    v.z = fmaf(v.z, r_sc.z, r_sh.z); v.w = fmaf(v.w, r_sc.w, r_sh.w);
                    if (r_islope != 1.f) {
                        v.x = v.x > 0.f ? v.x : v.x * r_islope; v.y = v.y > 0.f ? v.y : v.y * r_islope;
                        v.z = v.z > 0.f ? v.z : v.z * r_islope; v.w = v.w > 0.f ? v.w : v.w * r_islope;
                    }
                }
                amx_st4(s_x + (size_t)pix * SX + xg * 4, v);
            }
        }
        #pragma unroll
        for (int i = 0; i < DLD_MAX; ++i) {
            const int idx = tid + i * 256;
            if (idx < nd4) {
                const int pix = idx >> dg_shift, dg = idx & (DG - 1);
                float4 v = dr[i];
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

    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    int tile = blockIdx.x;
    // (tx, ty, n) of this workgroup's current tile, advanced by the split-K stride with carries (no division per tile)
    const int tpi = a.tiles_x * a.tiles_y;
    int ttx = tile % a.tiles_x, tty = (tile / a.tiles_x) % a.tiles_y, ttn = tile / tpi;
    const int stx = a.ksplit % a.tiles_x, sty = (a.ksplit / a.tiles_x) % a.tiles_y, stn = a.ksplit / tpi;
#ifdef AMX_WGRAD_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pl = __builtin_amdgcn_s_memtime();
    const unsigned long long pstart = pl;
#endif
    if (tile < ntiles) issue(ttn, tty, ttx);
    for (; tile < ntiles; tile += a.ksplit) {
        WG_TICK(2);
        stage();
        WG_TICK(0);
        __syncthreads();
        WG_TICK(1);
        if (tile + a.ksplit < ntiles) {
            ttx += stx; int carry = ttx >= a.tiles_x ? 1 : 0; ttx -= carry ? a.tiles_x : 0;
            tty += sty + carry; carry = tty >= a.tiles_y ? 1 : 0; tty -= carry ? a.tiles_y : 0;
            ttn += stn + carry;
            issue(ttn, tty, ttx);
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
                    const int dy = (TAPS == 9) ? (t / 3 - 1) * a.dil : 0;
                    const int dx = (TAPS == 9) ? (t % 3 - 1) * a.dil : 0;
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

template <int TAPS, int NT, int WM, int MAXHALO, int TH>
static int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    constexpr int CIB = 16 * WM;
    constexpr int SX = (CIB % 32 == 16) ? CIB : CIB + 16;
    const int halo = (TAPS == 9) ? a.dil : 0;
    const int COB = 16 * NT * a.WN;
    const int SD = (COB % 32 == 16) ? COB : COB + 16;
    const size_t lds = ((size_t)(TH + 2 * halo) * (TW + 2 * halo) * SX + (size_t)TH * TW * SD) * sizeof(float);
    dim3 grid(a.ksplit, amx_ceil_div(a.ci_pad, CIB) * a.co_blocks);
#ifndef AMX_EMU
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<TAPS, NT, WM, MAXHALO, TH>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
#endif
    AMX_LAUNCH((wgrad_kernel<TAPS, NT, WM, MAXHALO, TH>), grid, dim3(256), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

struct WgradPlan { int NT, WM, WN, WK, ksplit, rows, ci_pad, co_pad, th; };

static WgradPlan plan_wgrad(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    WgradPlan pl;
    pl.ci_pad = amx_round_up(Cin_s, 16);
    pl.co_pad = amx_round_up(cout, 16);
    pl.NT = pl.co_pad >= 32 ? 2 : 1;
    pl.WM = pl.ci_pad >= 64 ? 4 : (pl.ci_pad >= 32 ? 2 : 1);
    if (taps == 9 && dil > 1) pl.WM = 1;
    const int rem = 4 / pl.WM;
    const int co_tiles = pl.co_pad / (16 * pl.NT);                // wave-level co tiles needed
    pl.WN = 1;
    while (pl.WN * 2 <= rem && pl.WN * 2 <= co_tiles && 16 * pl.NT * pl.WN * 2 <= 64) pl.WN *= 2;
    pl.WK = rem / pl.WN;
    const int blocks = amx_ceil_div(pl.ci_pad, 16 * pl.WM) * amx_ceil_div(pl.co_pad, 16 * pl.NT * pl.WN);
    // Tile height / split-K target.  Stand-alone, layers with <= 32 input channels run 15-25 % faster with 4-row
    // tiles and two workgroups per CU (profiles/r01_wgrad_sweep.md), but inside the training step the weight
    // gradients run on the side stream next to the data-gradient convolutions, and there the lighter
    // one-workgroup-per-CU launch wins (20.79 vs 21.43 ms/step, interleaved A/B with AMX_WGRAD_LIGHT): the step
    // is what is optimised.  AMX_WGRAD_LIGHT=<wgs> switches the stand-alone optimum on for experiments.
    bool light = false;
    int light_wgs = 512;
    if (const char* e = getenv("AMX_WGRAD_LIGHT")) { const int v = atoi(e); if (v >= 64) { light = pl.ci_pad <= 32; light_wgs = v; } }
    pl.th = (taps == 9 && dil == 1 && (light || pl.NT == 2)) ? 4 : 8;
    if (const char* e = getenv("AMX_WGRAD_TH")) { const int v = atoi(e); if (v == 8 || (v == 4 && taps == 9 && dil == 1)) pl.th = v; }
    if (pl.WK > pl.th) pl.WK = pl.th;
    const int ntiles = amx_ceil_div(W, TW) * amx_ceil_div(H, pl.th) * N;
    // split-K workgroups: one per CU (256: 20.98 ms/step, 512: 21.5, 1024: 21.6, 128: 25.2)
    int target = light ? light_wgs : 256;
    if (const char* e = getenv("AMX_WGRAD_WGS")) { const int v = atoi(e); if (v >= 64) target = v; }
    int ks = amx_ceil_div(target, blocks);
    if (ks > ntiles) ks = ntiles;
    if (ks < 1) ks = 1;
    pl.ksplit = ks;
    pl.rows = ks * pl.WK;
    return pl;
}

extern "C" int amx_conv2d_wgrad_ksplit(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    return plan_wgrad(N, H, W, Cin_s, cout, taps, dil).ksplit;
}

// rows / floats of the partial buffer amx_conv2d_wgrad needs
extern "C" int amx_conv2d_wgrad_rows(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    return plan_wgrad(N, H, W, Cin_s, cout, taps, dil).rows;
}

static int wgrad_common(const float* x0, const float* sc0, const float* sh0, int C0s,
                        const float* x1, const float* sc1, const float* sh1, int C1s,
                        const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                        int taps, int dil, const float* aux, const float* k1, const float* k2, const float* k3,
                        float bslope, float* bpart, void* stream, float in_slope0 = 1.f, float in_slope1 = 1.f);

extern "C" int amx_conv2d_wgrad(const float* x0, const float* sc0, const float* sh0, int C0s,
                                const float* x1, const float* sc1, const float* sh1, int C1s,
                                const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                                int taps, int dil, void* stream) {
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dpre, Dos, part, N, H, W, cout, taps, dil, nullptr,
                        nullptr, nullptr, nullptr, 1.f, nullptr, stream);
}

// Weight gradient with the layer's BatchNorm/LeakyReLU backward fused into the dy loader and the bias gradient's
// partial sums (bpart: [amx_conv2d_wgrad_ksplit][round_up(cout,16)]) produced on the way.
extern "C" int amx_conv2d_wgrad_fused(const float* x0, const float* sc0, const float* sh0, int C0s,
                                      const float* x1, const float* sc1, const float* sh1, int C1s,
                                      const float* dy, const float* aux, const float* k1, const float* k2,
                                      const float* k3, float bslope, int Dos, float* part, float* bpart,
                                      int N, int H, int W, int cout, int taps, int dil, void* stream) {
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dy, Dos, part, N, H, W, cout, taps, dil, aux, k1, k2,
                        k3, bslope, bpart, stream);
}

// amx_conv2d_wgrad_fused for a layer whose input is read as LeakyReLU(affine(x)) (ResBlock's second conv)
extern "C" int amx_conv2d_wgrad_act(const float* x0, const float* sc0, const float* sh0, float in_slope0, int C0s,
                                    const float* x1, const float* sc1, const float* sh1, float in_slope1, int C1s,
                                    const float* dy, const float* aux, const float* k1, const float* k2,
                                    const float* k3, float bslope, int Dos, float* part, float* bpart,
                                    int N, int H, int W, int cout, int taps, int dil, void* stream) {
    if (!(in_slope0 > 0.f) || !(in_slope1 > 0.f)) AMX_BADARG(8);
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dy, Dos, part, N, H, W, cout, taps, dil, aux, k1, k2,
                        k3, bslope, bpart, stream, in_slope0, in_slope1);
}

extern "C" int amx_conv2d_wgrad_ksplit(int N, int H, int W, int Cin_s, int cout, int taps, int dil);

static int wgrad_common(const float* x0, const float* sc0, const float* sh0, int C0s,
                        const float* x1, const float* sc1, const float* sh1, int C1s,
                        const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                        int taps, int dil, const float* aux, const float* k1, const float* k2, const float* k3,
                        float bslope, float* bpart, void* stream, float in_slope0, float in_slope1) {
    if (!x0 || !dpre || !part) AMX_BADARG(1);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(7);
    if (N <= 0 || H <= 0 || W <= 0 || cout <= 0) AMX_BADARG(2);
    if ((C0s & 3) || (C1s & 3) || (Dos & 3) || C0s <= 0 || Dos < cout) AMX_BADARG(3);
    if (taps != 1 && taps != 9) AMX_BADARG(4);
    if (taps == 9 && (dil < 1 || dil > 6)) AMX_BADARG(5);
    if ((x1 == nullptr) != (C1s == 0)) AMX_BADARG(6);
    const WgradPlan pl = plan_wgrad(N, H, W, C0s + C1s, cout, taps, dil);
    WgradArgs a;
    a.x0 = x0; a.sc0 = sc0; a.sh0 = sh0; a.C0s = C0s;
    a.x1 = x1; a.sc1 = sc1; a.sh1 = sh1; a.C1s = C1s;
    a.in_slope0 = in_slope0; a.in_slope1 = in_slope1;
    a.dpre = dpre; a.Dos = Dos; a.part = part;
    a.prof = nullptr;
#ifdef AMX_WGRAD_PROFILE
    a.prof = (unsigned long long*)amx_wgrad_profile_buffer;
#endif
    a.aux = aux; a.k1 = k1; a.k2 = k2; a.k3 = k3; a.bslope = bslope; a.bpart = bpart;
    a.N = N; a.H = H; a.W = W; a.dil = dil;
    a.ci_pad = pl.ci_pad; a.co_pad = pl.co_pad;
    a.WN = pl.WN; a.WK = pl.WK; a.ksplit = pl.ksplit;
    a.tiles_x = amx_ceil_div(W, TW); a.tiles_y = amx_ceil_div(H, pl.th);
    a.co_blocks = amx_ceil_div(pl.co_pad, 16 * pl.NT * pl.WN);
    hipStream_t s = (hipStream_t)stream;
#define WG_DISPATCH(T, H_, TH_)                                                   \
    if (pl.NT == 1) {                                                              \
        if (pl.WM == 1) return launch_wgrad<T, 1, 1, H_, TH_>(a, s);                \
        if (pl.WM == 2) return launch_wgrad<T, 1, 2, H_, TH_>(a, s);                \
        return launch_wgrad<T, 1, 4, H_, TH_>(a, s);                                \
    } else {                                                                       \
        if (pl.WM == 1) return launch_wgrad<T, 2, 1, H_, TH_>(a, s);                \
        if (pl.WM == 2) return launch_wgrad<T, 2, 2, H_, TH_>(a, s);                \
        return launch_wgrad<T, 2, 4, H_, TH_>(a, s);                                \
    }
    if (taps == 1) { WG_DISPATCH(1, 0, 8) }
    if (dil == 1) {
        if (pl.th == 4) { WG_DISPATCH(9, 1, 4) }
        WG_DISPATCH(9, 1, 8)
    }
    if (pl.NT == 1) return launch_wgrad<9, 1, 1, 6, 8>(a, s);
    return launch_wgrad<9, 2, 1, 6, 8>(a, s);
#undef WG_DISPATCH
}
