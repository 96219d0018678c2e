// conv1.hip — first-layer convolution (Cin == 1: the microscopy image itself) and its weight gradient.
// K = 9 is too small for MFMA; the layer is bound by the HBM write of its Cout-channel output, so this is
// a VALU kernel organised for perfectly coalesced 16 B/lane NHWC stores: G = Cs/4 lanes share a pixel,
// each producing 4 output channels.
//
//   c1 = ConvBlock(2, nbl[0], 1, nb_filters): Conv2d(1, F, 3, padding=1) -> LeakyReLU -> BN stats
//                                                   atomai/nets/fcnn.py:66-69, 186-189; blocks.py:61-76
#include "amx_device.h"
#ifndef AMX_CONV1_WINDOW_POOL
#define AMX_CONV1_WINDOW_POOL 1  // compile-time A/B switch: 0 = the round-3 form (pixel loop + a second loop recomputing the windows)
#endif
#include <cstdlib>
#ifndef AMX_CONV1_UNROLL
#define AMX_CONV1_UNROLL 4      // weight gradient: pixels of a thread in flight together (a thread walks rows_pix / PL
                                // pixels, each a global-load round trip): 192 -> 160 us per launch at bs 32, 512^2
#endif
#ifndef AMX_CONV1_UNROLL_FWD
#define AMX_CONV1_UNROLL_FWD 1  // forward: 4 in flight cost occupancy (131 VGPRs) and measured slower (207 -> 242 us)
#endif
#ifndef AMX_CONV1_FAST
#define AMX_CONV1_FAST 1        // compile-time experiment switch: interior pixels skip the per-tap bounds arithmetic
#endif

#ifndef AMX_CONV1_LDS
#define AMX_CONV1_LDS 1         // the block's pixel range + one halo row either side staged in LDS (see stage_image)
#endif
#define CONV1_STAGE_MAX 12288   // floats (48 KB): wider images fall back to direct global loads

// Per-thread shifted sums for 4 channels: d = x - K with K = the first value the thread sees, so that
// M2 = sum d^2 - (sum d)^2 / n is free of catastrophic cancellation; merged across the block with Chan's formula.
struct Sh4 { float4 K, s1, s2; float n; };

__device__ __forceinline__ void sh_push(Sh4& s, const float4 v) {
    if (s.n == 0.f) s.K = v;
    s.n += 1.f;
    float d;
    d = v.x - s.K.x; s.s1.x += d; s.s2.x = fmaf(d, d, s.s2.x);
    d = v.y - s.K.y; s.s1.y += d; s.s2.y = fmaf(d, d, s.s2.y);
    d = v.z - s.K.z; s.s1.z += d; s.s2.z = fmaf(d, d, s.s2.z);
    d = v.w - s.K.w; s.s1.w += d; s.s2.w = fmaf(d, d, s.s2.w);
}

// Pixel cursor of a thread that walks p, p + step, p + 2*step, ...: one 64-bit division at the start, then
// incremental (x, y, image) updates (the per-pixel `p % W`, `p / W % H` cost more than the 36 FMAs of the tap loop).
struct PixCursor {
    int xx, yy; long nimg;
    __device__ __forceinline__ void init(long p, int H, int W) {
        xx = (int)(p % W); const long r = p / W; yy = (int)(r % H); nimg = r / H;
    }
    __device__ __forceinline__ void advance(int step, int H, int W) {
        xx += step;
        while (xx >= W) { xx -= W; if (++yy == H) { yy = 0; ++nimg; } }
    }
};

// The 3x3 neighbourhood of pixel (yy, xx) of a single-channel image, zero padded — BRANCH-FREE: row / column indices
// are clamped into the image (every load is legal), out-of-image taps are zeroed by selects afterwards.  The round-1/2
// form (an interior fast path, per-tap branches on the border) compiled to a load + s_waitcnt inside every branch, so a
// thread had ONE global load in flight at a time and the unrolled pixel loop could not overlap pixels either: the two
// first-layer kernels ran at 2.8-2.9 TB/s, latency-bound.  With straight-line code the 9 x AMX_CONV1_UNROLL loads of a
// thread are issued back to back.
// in_sub / in_div: the predictor's stack normalisation (x - min) / ptp (utils/preproc.py:822-823) applied to in-bounds
// values (the padding stays zero); (0, 1) is the exact identity.
template <bool FAST>
static __device__ __forceinline__ void load_3x3(const float* __restrict__ img, int yy, int xx, int H, int W, int dil,
                                                float v[9], bool norm = false, float in_sub = 0.f, float in_div = 1.f) {
    const int ym = yy - dil, yp = yy + dil, xm = xx - dil, xp = xx + dil;
    const bool oky[3] = {ym >= 0, true, yp < H}, okx[3] = {xm >= 0, true, xp < W};
    const float* r[3] = {img + (size_t)(ym >= 0 ? ym : 0) * W, img + (size_t)yy * W, img + (size_t)(yp < H ? yp : H - 1) * W};
    const int c[3] = {xm >= 0 ? xm : 0, xx, xp < W ? xp : W - 1};
    #pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = r[t / 3][c[t % 3]];
    #pragma unroll
    for (int t = 0; t < 9; ++t) {
        float u = v[t];
        if (norm) { const float d = u - in_sub; u = d / in_div; }
        v[t] = (oky[t / 3] && okx[t % 3]) ? u : 0.f;
    }
}

// LDS staging of the single-channel input (round 3).  A block owns the LINEAR pixel range [p0, p1); every tap of every
// pixel of that range lies in [p0 - halo, p1 + halo) with halo = dil * (W + 1), so the block copies that range once with
// coalesced loads (positions outside the tensor read as 0; row / image borders are still masked per tap by the caller)
// and the pixel loop reads its 9 taps from LDS.  The direct form issued 9 dependent global loads per pixel and thread
// and ran at 2.6-2.8 TB/s of output, latency-bound.  Values and FMA order are unchanged.
// Measured (profiles/r03_conv1_lds_ab.log, in-process): U-Net 32 x 512^2 x 16 forward 238 -> 174 us; dilnet predict
// 16 x 1024^2 x 28 (no statistics, input normalisation on load) 818 -> 426 us = 4.4 TB/s of output.  The weight-gradient
// kernel, which already keeps 4 pixels (36 loads) in flight per thread, got SLOWER with the same staging (158 -> 175 us)
// and stays on direct loads.
static __device__ __forceinline__ void stage_image(float* s_in, const float* __restrict__ x, long p0, long npix, int halo,
                                                   int len, bool norm, float in_sub, float in_div) {
    for (int i = threadIdx.x; i < len; i += 256) {
        const long q = p0 - halo + i;
        float v = (q >= 0 && q < npix) ? x[q] : 0.f;
        if (norm) { const float d = v - in_sub; v = d / in_div; }
        s_in[i] = v;
    }
    __syncthreads();
}
static __device__ __forceinline__ void taps_lds(const float* s_in, int o, int yy, int xx, int H, int W, int dil, float v[9]) {
    const bool oky[3] = {yy - dil >= 0, true, yy + dil < H}, okx[3] = {xx - dil >= 0, true, xx + dil < W};
    #pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float u = s_in[o + (t / 3 - 1) * W * dil + (t % 3 - 1) * dil];
        v[t] = (oky[t / 3] && okx[t % 3]) ? u : 0.f;
    }
}

// x [N][H][W] (single channel), w OIHW [Cout][1][3][3], y NHWC [P][Cs].
// Block b owns pixels [b*ppb, (b+1)*ppb); stats row b = (sum, M2 about the row mean) per channel.
template <bool STAGED, bool POOL>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, float* __restrict__ stats,
                                                        int N, int H, int W, int Cout, int Cs, int dil,
                                                        float slope, int ppb, int cop, float in_sub, float in_div,
                                                        int stage_len, float* __restrict__ pool_out,
                                                        const float* __restrict__ pscale, const float* __restrict__ pshift) {
    const int G = Cs >> 2, PL = 256 / G;
    const int tid = threadIdx.x;
    const int pl = tid / G, cg = tid - pl * G;
    const bool active = pl < PL;
    AMX_DYN_SMEM(float, s);                       // [PL][3][Cs]: mean, m2, n
    const long npix = (long)N * H * W;
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    float4 wt[9];
    float4 b4 = make_float4(0, 0, 0, 0);
    if (active) {
        #pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int c = cg * 4;
            wt[t].x = c + 0 < Cout ? w[(c + 0) * 9 + t] : 0.f; wt[t].y = c + 1 < Cout ? w[(c + 1) * 9 + t] : 0.f;
            wt[t].z = c + 2 < Cout ? w[(c + 2) * 9 + t] : 0.f; wt[t].w = c + 3 < Cout ? w[(c + 3) * 9 + t] : 0.f;
        }
        const int c = cg * 4;
        if (bias) {
            b4.x = c + 0 < Cout ? bias[c + 0] : 0.f; b4.y = c + 1 < Cout ? bias[c + 1] : 0.f;
            b4.z = c + 2 < Cout ? bias[c + 2] : 0.f; b4.w = c + 3 < Cout ? bias[c + 3] : 0.f;
        }
    }
    const bool norm = !(in_sub == 0.f && in_div == 1.f);     // uniform: training never normalises here
    Sh4 st; st.K = make_float4(0, 0, 0, 0); st.s1 = st.K; st.s2 = st.K; st.n = 0.f;
    const int halo = dil * (W + 1);
    if (STAGED) stage_image(s, x, p0, npix, halo, stage_len, norm, in_sub, in_div);      // (aliases the statistics rows)
    // Eval mode with the 2x2 max-pool fused in (POOL), dilation 1 — round 5: a thread owns one 2 x 2 WINDOW instead of one
    // pixel: the four outputs share a 4 x 4 patch of the staged image (16 LDS reads instead of 4 x 9), are written as two
    // pairs of adjacent pixels, and their affine + maximum is the pooled output — no second loop that recomputes the four
    // convolutions (it cost more than the bytes it wrote: 720 vs 430 us per 16 dilnet frames with / without the pooled
    // tensor).  Per output the FMA order over the taps is unchanged and the maximum is taken in pool_fwd_kernel's order, so
    // y and pool_out are bit-identical to the separate launches.
    const bool window_path = AMX_CONV1_WINDOW_POOL && STAGED && POOL && dil == 1;
    if (window_path && active) {
        const int Wo = W >> 1;
        const long row0 = p0 / W;                            // global row (n * H + y) of the block's first row: even
        const int npool = (int)((p1 - p0) / W >> 1) * Wo;
        const float4 sc = amx_ld4(pscale + cg * 4), sh = amx_ld4(pshift + cg * 4);
        for (int j = pl; j < npool; j += PL) {
            const int pr = j / Wo, px = j - pr * Wo;
            const long grow = row0 + 2 * pr;
            const int yy = (int)(grow % H), xx = 2 * px;
            const int o = 2 * pr * W + xx + halo;              // staged-image offset of the window's top-left pixel
            float pv[4][4];                                   // rows yy - 1 .. yy + 2, columns xx - 1 .. xx + 2 (masked)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool oky = (yy + r - 1 >= 0) && (yy + r - 1 < H);
                #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool okx = (xx + c - 1 >= 0) && (xx + c - 1 < W);
                    const float u = s[o + (r - 1) * W + (c - 1)];
                    pv[r][c] = (oky && okx) ? u : 0.f;
                }
            }
            float4 best = make_float4(0, 0, 0, 0);
            #pragma unroll
            for (int k = 0; k < 4; ++k) {                     // window pixels in pool_fwd_kernel's order: (0,0) (0,1) (1,0) (1,1)
                const int ky = k >> 1, kx = k & 1;
                float4 acc = b4;
                #pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = pv[ky + t / 3][kx + t % 3];
                    acc.x = fmaf(v, wt[t].x, acc.x); acc.y = fmaf(v, wt[t].y, acc.y);
                    acc.z = fmaf(v, wt[t].z, acc.z); acc.w = fmaf(v, wt[t].w, acc.w);
                }
                acc.x = acc.x > 0.f ? acc.x : acc.x * slope; acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
                acc.z = acc.z > 0.f ? acc.z : acc.z * slope; acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
                amx_st4(y + (size_t)(p0 + (long)(2 * pr + ky) * W + xx + kx) * Cs + cg * 4, acc);
                acc.x = fmaf(acc.x, sc.x, sh.x); acc.y = fmaf(acc.y, sc.y, sh.y);
                acc.z = fmaf(acc.z, sc.z, sh.z); acc.w = fmaf(acc.w, sc.w, sh.w);
                if (k == 0) best = acc;
                else {
                    best.x = acc.x > best.x ? acc.x : best.x; best.y = acc.y > best.y ? acc.y : best.y;
                    best.z = acc.z > best.z ? acc.z : best.z; best.w = acc.w > best.w ? acc.w : best.w;
                }
            }
            amx_st4(pool_out + ((size_t)(grow >> 1) * Wo + px) * Cs + cg * 4, best);
        }
    }
    if (active && !window_path)
    {
        PixCursor cur; cur.init(p0 + pl < npix ? p0 + pl : 0, H, W);
        constexpr int U = AMX_CONV1_UNROLL_FWD;
        for (long p = p0 + pl; p < p1; p += (long)PL * U) {
            // U pixels of this thread in flight: all their neighbourhood loads are issued before the first FMA
            float xv[U][9];
            bool ok[U];
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = p + (long)u * PL < p1;
                const long ni = ok[u] ? cur.nimg : 0;
                if (STAGED) taps_lds(s, (int)((ok[u] ? p + (long)u * PL : p0) - p0) + halo, ok[u] ? cur.yy : 0, ok[u] ? cur.xx : 0, H, W, dil, xv[u]);
                else load_3x3<true>(x + (size_t)ni * H * W, ok[u] ? cur.yy : 0, ok[u] ? cur.xx : 0, H, W, dil, xv[u], norm,
                                    in_sub, in_div);
                if (ok[u]) cur.advance(PL, H, W);
            }
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                float4 acc = b4;
                #pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = xv[u][t];
                    acc.x = fmaf(v, wt[t].x, acc.x); acc.y = fmaf(v, wt[t].y, acc.y);
                    acc.z = fmaf(v, wt[t].z, acc.z); acc.w = fmaf(v, wt[t].w, acc.w);
                }
                acc.x = acc.x > 0.f ? acc.x : acc.x * slope; acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
                acc.z = acc.z > 0.f ? acc.z : acc.z * slope; acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
                amx_st4(y + (size_t)(p + (long)u * PL) * Cs + cg * 4, acc);
                sh_push(st, acc);
            }
        }
    }
    // Eval mode, block followed by F.max_pool2d(x, 2, 2) (fcnn.py:123, 219): the pooled, NORMALISED tensor is produced here
    // as well — the four convolution outputs of a window are recomputed from the staged image (16 LDS reads, 144 FMAs per
    // 4 channels: the kernel is bound by its stores, not by the VALU), the BatchNorm affine is applied before the max and the
    // maximum is taken in pool_fwd_kernel's order, so the result is bit-identical to amx_pool2x2_fwd of y.  Saves reading
    // y back (the largest activation of the net) and a launch.  The launcher guarantees whole row pairs per block.
    if (STAGED && POOL && active && !window_path) {
        const int Wo = W >> 1;
        const long row0 = p0 / W;                            // global row (n * H + y) of the block's first row: even
        const int npool = (int)((p1 - p0) / W >> 1) * Wo;
        const float4 sc = amx_ld4(pscale + cg * 4), sh = amx_ld4(pshift + cg * 4);
        for (int j = pl; j < npool; j += PL) {
            const int pr = j / Wo, px = j - pr * Wo;
            const long grow = row0 + 2 * pr;
            const int yy = (int)(grow % H);
            float4 best = make_float4(0, 0, 0, 0);
            #pragma unroll
            for (int k = 0; k < 4; ++k) {
                float xv[9];
                taps_lds(s, (2 * pr + (k >> 1)) * W + 2 * px + (k & 1) + halo, yy + (k >> 1), 2 * px + (k & 1), H, W, dil, xv);
                float4 acc = b4;
                #pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = xv[t];
                    acc.x = fmaf(v, wt[t].x, acc.x); acc.y = fmaf(v, wt[t].y, acc.y);
                    acc.z = fmaf(v, wt[t].z, acc.z); acc.w = fmaf(v, wt[t].w, acc.w);
                }
                acc.x = acc.x > 0.f ? acc.x : acc.x * slope; acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
                acc.z = acc.z > 0.f ? acc.z : acc.z * slope; acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
                acc.x = fmaf(acc.x, sc.x, sh.x); acc.y = fmaf(acc.y, sc.y, sh.y);
                acc.z = fmaf(acc.z, sc.z, sh.z); acc.w = fmaf(acc.w, sc.w, sh.w);
                if (k == 0) best = acc;
                else {
                    best.x = acc.x > best.x ? acc.x : best.x; best.y = acc.y > best.y ? acc.y : best.y;
                    best.z = acc.z > best.z ? acc.z : best.z; best.w = acc.w > best.w ? acc.w : best.w;
                }
            }
            amx_st4(pool_out + ((size_t)(grow >> 1) * Wo + px) * Cs + cg * 4, best);
        }
    }
    if (!stats) return;
    if (STAGED) __syncthreads();                  // every thread is done with the staged image: its LDS becomes the rows below
    // Block statistics: Chan merges of the PL per-thread (mean, M2, n) triples of every channel group, as a binary
    // tree over the pixel lanes (fixed order, 6-8 rounds on all lanes).  The round-1/2 form walked the PL rows serially
    // on Cs threads while the other 240 waited: ~40 % of a training-mode block's life.
    float4 mean = make_float4(0, 0, 0, 0), m2 = mean;
    float n = 0.f;
    if (active && st.n > 0.f) {
        const float inv = 1.f / st.n;
        n = st.n;
        mean.x = st.K.x + st.s1.x * inv; m2.x = st.s2.x - st.s1.x * st.s1.x * inv;
        mean.y = st.K.y + st.s1.y * inv; m2.y = st.s2.y - st.s1.y * st.s1.y * inv;
        mean.z = st.K.z + st.s1.z * inv; m2.z = st.s2.z - st.s1.z * st.s1.z * inv;
        mean.w = st.K.w + st.s1.w * inv; m2.w = st.s2.w - st.s1.w * st.s1.w * inv;
    }
    int span = 1;
    while (span < PL) span <<= 1;
    for (int half = span >> 1; half >= 1; half >>= 1) {
        if (active && pl >= half && pl < 2 * half) {       // rows [half, 2 half) hand over to rows [0, half)
            amx_st4(s + ((size_t)(pl * 3 + 0) * Cs + cg * 4), mean);
            amx_st4(s + ((size_t)(pl * 3 + 1) * Cs + cg * 4), m2);
            s[(size_t)(pl * 3 + 2) * Cs + cg * 4] = n;
        }
        __syncthreads();
        if (active && pl < half && pl + half < PL) {
            const int q = pl + half;
            const float nq = s[(size_t)(q * 3 + 2) * Cs + cg * 4];
            if (nq > 0.f) {
                const float4 mq = amx_ld4(s + ((size_t)(q * 3 + 0) * Cs + cg * 4));
                const float4 m2q = amx_ld4(s + ((size_t)(q * 3 + 1) * Cs + cg * 4));
                const float nt = n + nq, f = nq / nt, g2 = n * nq / nt;
                float d;
                d = mq.x - mean.x; mean.x += d * f; m2.x += m2q.x + d * d * g2;
                d = mq.y - mean.y; mean.y += d * f; m2.y += m2q.y + d * d * g2;
                d = mq.z - mean.z; mean.z += d * f; m2.z += m2q.z + d * d * g2;
                d = mq.w - mean.w; mean.w += d * f; m2.w += m2q.w + d * d * g2;
                n = nt;
            }
        }
        __syncthreads();
    }
    if (active && pl == 0) {
        amx_st4(stats + ((size_t)blockIdx.x * 2 + 0) * cop + cg * 4, make_float4(mean.x * n, mean.y * n, mean.z * n, mean.w * n));
        amx_st4(stats + ((size_t)blockIdx.x * 2 + 1) * cop + cg * 4, m2);
    }
}

// floats of the staged range of a block (0 = direct loads: switched off, or the range does not fit)
static int conv1_stage_len(int ppb, int W, int dil) {
    if (!AMX_CONV1_LDS) return 0;
    const long len = (long)ppb + 2L * dil * (W + 1);
    return len <= CONV1_STAGE_MAX ? (int)len : 0;
}

static int conv1_fwd_common(const float* x, const float* w, const float* bias, float* y, float* stats,
                            int N, int H, int W, int Cout, int Cs, int dil, float slope, int rows,
                            int rows_pix, float in_sub, float in_div, float* pool_out, const float* pscale,
                            const float* pshift, void* stream) {
    if (!x || !w || !y || Cout <= 0 || Cs < Cout || (Cs & 3) || Cs > 256 || dil < 1) AMX_BADARG(1);
    if (!(in_div != 0.f)) AMX_BADARG(3);
    const long npix = (long)N * H * W;
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(2);
    const int PL = 256 / (Cs / 4);
    const int stage_len = conv1_stage_len(rows_pix, W, dil);
    if (pool_out && (!stage_len || stats || !pscale || !pshift || (H & 1) || (W & 1) || rows_pix % (2 * W))) AMX_BADARG(4);
    size_t lds = (size_t)PL * 3 * Cs * sizeof(float);
    if ((size_t)stage_len * sizeof(float) > lds) lds = (size_t)stage_len * sizeof(float);
    if (stage_len && pool_out)
        AMX_LAUNCH((conv1_fwd_kernel<true, true>), dim3(rows), dim3(256), lds,
                   (hipStream_t)stream, x, w, bias, y, stats, N, H, W, Cout, Cs, dil, slope, rows_pix,
                   amx_round_up(Cout, 16), in_sub, in_div, stage_len, pool_out, pscale, pshift);
    else if (stage_len)
        AMX_LAUNCH((conv1_fwd_kernel<true, false>), dim3(rows), dim3(256), lds,
                   (hipStream_t)stream, x, w, bias, y, stats, N, H, W, Cout, Cs, dil, slope, rows_pix,
                   amx_round_up(Cout, 16), in_sub, in_div, stage_len, pool_out, pscale, pshift);
    else
        AMX_LAUNCH((conv1_fwd_kernel<false, false>), dim3(rows), dim3(256), lds,
                   (hipStream_t)stream, x, w, bias, y, stats, N, H, W, Cout, Cs, dil, slope, rows_pix,
                   amx_round_up(Cout, 16), in_sub, in_div, stage_len, (float*)nullptr, (const float*)nullptr,
                   (const float*)nullptr);
    AMX_CHECK_LAUNCH();
    return 0;
}

extern "C" int amx_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* stats,
                             int N, int H, int W, int Cout, int Cs, int dil, float slope, int rows,
                             int rows_pix, float in_sub, float in_div, void* stream) {
    return conv1_fwd_common(x, w, bias, y, stats, N, H, W, Cout, Cs, dil, slope, rows, rows_pix, in_sub, in_div, nullptr,
                            nullptr, nullptr, stream);
}

// 1 when amx_conv1_fwd_pool can serve this shape (even H and W, whole row pairs per block, the block's range fits the LDS)
extern "C" int amx_conv1_fwd_pool_supported(int H, int W, int dil, int rows_pix) {
    return conv1_stage_len(rows_pix, W, dil) > 0 && !(H & 1) && !(W & 1) && rows_pix % (2 * W) == 0;
}

// Eval-mode first layer followed by a 2x2 max-pool: y as amx_conv1_fwd (no statistics) AND pooled =
// max_pool2d(y * pscale + pshift) [N][H/2][W/2][Cs], bit-identical to amx_pool2x2_fwd(y, pscale, pshift).
extern "C" int amx_conv1_fwd_pool(const float* x, const float* w, const float* bias, float* y, float* pooled,
                                  const float* pscale, const float* pshift, int N, int H, int W, int Cout, int Cs,
                                  int dil, float slope, int rows, int rows_pix, float in_sub, float in_div, void* stream) {
    if (!pooled) AMX_BADARG(5);
    return conv1_fwd_common(x, w, bias, y, nullptr, N, H, W, Cout, Cs, dil, slope, rows, rows_pix, in_sub, in_div, pooled,
                            pscale, pshift, stream);
}

// dW[co][0][t] = sum_p dpre[p][co] * x[p + tap t];  partial rows part[blk][9][Cs] (+ row 9 = sum_p dpre when
// aux != nullptr: dpre = lrelu'(a) * (k1*dy + k2*a + k3) is then formed on load and the bias gradient comes along)
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dpre,
                                                          const float* __restrict__ aux,
                                                          const float* __restrict__ k1,
                                                          const float* __restrict__ k2,
                                                          const float* __restrict__ k3, float bslope,
                                                          float* __restrict__ part, int N, int H, int W,
                                                          int Cs, int dil, int ppb, int nrow) {
    const int G = Cs >> 2, PL = 256 / G;
    const int tid = threadIdx.x;
    const int pl = tid / G, cg = tid - pl * G;
    const bool active = pl < PL;
    AMX_DYN_SMEM(float, s);                       // [nrow][PL][Cs]
    const long npix = (long)N * H * W;
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    float4 acc[10];
    #pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = make_float4(0, 0, 0, 0);
    float4 c1 = make_float4(1, 1, 1, 1), c2 = make_float4(0, 0, 0, 0), c3 = c2;
    if (active && aux && k1) { c1 = amx_ld4(k1 + cg * 4); c2 = amx_ld4(k2 + cg * 4); c3 = amx_ld4(k3 + cg * 4); }
    if (active)
    {
        PixCursor cur; cur.init(p0 + pl < npix ? p0 + pl : 0, H, W);
        constexpr int U = AMX_CONV1_UNROLL;
        for (long p = p0 + pl; p < p1; p += (long)PL * U) {
            float xv[U][9];
            float4 gq[U], tq[U];
            bool ok[U];
            #pragma unroll
            for (int u = 0; u < U; ++u) {                    // all loads of U pixels first
                ok[u] = p + (long)u * PL < p1;
                const long pu = ok[u] ? p + (long)u * PL : p;
                gq[u] = amx_ld4(dpre + (size_t)pu * Cs + cg * 4);
                tq[u] = aux ? amx_ld4(aux + (size_t)pu * Cs + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const long ni = ok[u] ? cur.nimg : 0;
                load_3x3<false>(x + (size_t)ni * H * W, ok[u] ? cur.yy : 0, ok[u] ? cur.xx : 0, H, W, dil, xv[u]);
                if (ok[u]) cur.advance(PL, H, W);
            }
            #pragma unroll
            for (int u = 0; u < U; ++u) {                    // accumulation in pixel order (as with one pixel at a time)
                if (!ok[u]) continue;
                float4 g = gq[u];
                if (aux) {
                    const float4 t = tq[u];
                    g.x = (t.x > 0.f ? 1.f : bslope) * fmaf(c1.x, g.x, fmaf(c2.x, t.x, c3.x));
                    g.y = (t.y > 0.f ? 1.f : bslope) * fmaf(c1.y, g.y, fmaf(c2.y, t.y, c3.y));
                    g.z = (t.z > 0.f ? 1.f : bslope) * fmaf(c1.z, g.z, fmaf(c2.z, t.z, c3.z));
                    g.w = (t.w > 0.f ? 1.f : bslope) * fmaf(c1.w, g.w, fmaf(c2.w, t.w, c3.w));
                }
                acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
                #pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = xv[u][t];
                    acc[t].x = fmaf(v, g.x, acc[t].x); acc[t].y = fmaf(v, g.y, acc[t].y);
                    acc[t].z = fmaf(v, g.z, acc[t].z); acc[t].w = fmaf(v, g.w, acc[t].w);
                }
            }
        }
    }
    // all tap rows (+ the bias row) of every pixel lane go to LDS at once, then nrow * Cs threads each sum one column
    // over the PL lanes in lane order (the same order, hence the same bits, as the round-1/2 form, which did this one
    // tap at a time on Cs threads with two barriers per tap: 10 serial rounds of a 64-step sum were longer than the
    // block's pixel loop)
    #pragma unroll
    for (int t = 0; t < 10; ++t)
        if (t < nrow && active) amx_st4(s + ((size_t)(t * PL + pl) * Cs + cg * 4), acc[t]);
    __syncthreads();
    for (int i = tid; i < nrow * Cs; i += 256) {
        const int t = i / Cs, c = i - t * Cs;
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += s[(size_t)(t * PL + q) * Cs + c];
        part[((size_t)blockIdx.x * nrow + t) * Cs + c] = a;
    }
}

extern "C" int amx_conv1_wgrad(const float* x, const float* dpre, float* part, int N, int H, int W,
                               int Cs, int dil, int rows, int rows_pix, void* stream) {
    if (!x || !dpre || !part || (Cs & 3) || Cs <= 0 || Cs > 256 || dil < 1) AMX_BADARG(1);
    const long npix = (long)N * H * W;
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(2);
    const int PL = 256 / (Cs / 4);
    AMX_LAUNCH(conv1_wgrad_kernel, dim3(rows), dim3(256), (size_t)9 * PL * Cs * sizeof(float),
               (hipStream_t)stream, x, dpre, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
               (const float*)nullptr, 1.f, part, N, H, W, Cs, dil, rows_pix, 9);
    AMX_CHECK_LAUNCH();
    return 0;
}

// fused variant: part is [rows][10][Cs]; tap rows 0..8 as above, row 9 = sum of dpre (bias gradient)
extern "C" int amx_conv1_wgrad_fused(const float* x, const float* dy, const float* aux, const float* k1,
                                     const float* k2, const float* k3, float bslope, float* part, int N, int H,
                                     int W, int Cs, int dil, int rows, int rows_pix, void* stream) {
    if (!x || !dy || !aux || !part || (Cs & 3) || Cs <= 0 || Cs > 256 || dil < 1) AMX_BADARG(1);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(2);
    const long npix = (long)N * H * W;
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(3);
    const int PL = 256 / (Cs / 4);
    AMX_LAUNCH(conv1_wgrad_kernel, dim3(rows), dim3(256), (size_t)10 * PL * Cs * sizeof(float),
               (hipStream_t)stream, x, dy, aux, k1, k2, k3, bslope, part, N, H, W, Cs, dil, rows_pix, 10);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Reduces wgrad partial rows [rows][taps][ci_pad][co_pad] into the OIHW gradient dW[co][ci][tap].
// A block owns 32 consecutive (tap, pci, co) columns (co fastest -> coalesced 128 B row reads) and
// splits the rows over 8 lanes; fp64 accumulation, fixed order -> deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int rows,
                                                           int taps, int ci_pad, int co_pad, int C0,
                                                           int C0s, int C1s, int Cin, int Cout,
                                                           float* __restrict__ dw) {
    __shared__ double red[8][32];
    const int tid = threadIdx.x;
    const int col = tid & 31, rl = tid >> 5;
    const long ncols = (long)taps * ci_pad * co_pad;
    const long c = (long)blockIdx.x * 32 + col;
    double acc = 0.0;
    if (c < ncols)
        for (int r = rl; r < rows; r += 8) acc += (double)part[(size_t)r * ncols + c];
    red[rl][col] = acc;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        double s = 0.0;
        #pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][col];
        const int co = (int)(c % co_pad);
        const long r2 = c / co_pad;
        const int pci = (int)(r2 % ci_pad);
        const int t = (int)(r2 / ci_pad);
        // padded-concat channel index -> real input channel (or none)
        int ci = -1;
        if (pci < C0s) { if (pci < C0) ci = pci; }
        else if (pci - C0s < C1s && pci - C0s < Cin - C0) ci = C0 + (pci - C0s);
        if (ci >= 0 && co < Cout) dw[((size_t)co * Cin + ci) * taps + t] = (float)s;
    }
}

extern "C" int amx_wgrad_reduce(const float* part, int rows, int taps, int ci_pad, int co_pad, int C0,
                                int C0s, int C1, int Cout, float* dw, void* stream) {
    if (!part || !dw || rows <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0 || co_pad < Cout) AMX_BADARG(1);
    if (C0s < C0) AMX_BADARG(2);
    const int Cin = C0 + C1;
    const int C1s = amx_round_up(C1, 4);
    const long ncols = (long)taps * ci_pad * co_pad;
    AMX_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((ncols + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
               part, rows, taps, ci_pad, co_pad, C0, C0s, C1s, Cin, Cout, dw);
    AMX_CHECK_LAUNCH();
    return 0;
}
