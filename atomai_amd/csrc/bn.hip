// bn.hip — BatchNorm2d (after LeakyReLU, reference order conv -> lrelu -> BN, atomai/nets/blocks.py:61-76)
// split into the pieces the fused pipeline needs.  All HBM-bound; NHWC float4 accesses.
//
//   amx_bn_finalize        partial (sum, M2) rows from the conv epilogue -> batch mean / biased var
//                          (fp64 Chan merge) -> scale = g*invstd, shift = b - mean*scale for the
//                          CONSUMER to apply on load; running stats updated with torch semantics
//                          (momentum 0.1, unbiased running_var).
//   amx_bn_eval_affine     eval mode: scale/shift from running statistics.
//   amx_affine_nhwc        materialises y = a*scale + shift (module-boundary outputs only).
//   amx_bn_bwd_reduce      per-channel sum(dy), sum(dy*a) partial rows.
//   amx_bn_bwd_finalize    -> dgamma, dbeta and the three per-channel constants of
//                          da = k1*dy + k2*a + k3  (batch-norm backward is affine in dy and a).
//   amx_bn_bwd_apply       dpre = lrelu'(a) * (k1*dy + k2*a + k3) [+ extra terms for DilatedBlock],
//                          plus per-channel partial sums of dpre (= conv bias gradient).
//   amx_reduce_rows        deterministic column sum of partial rows (fp64 accumulate).
#include "amx_device.h"

// Thread mapping shared by the per-channel reductions over an NHWC tensor: G = Cs/4 channel groups,
// PL = 256/G pixel lanes.  thread -> (pl = tid / G, cg = tid % G), idle if pl >= PL.
struct PixMap {
    int G, PL, pl, cg;
    bool active;
    __device__ PixMap(int Cs, int tid) {
        G = Cs >> 2; PL = 256 / G; pl = tid / G; cg = tid - pl * G; active = pl < PL;
    }
};

// ------------------------------------------------------------------ finalize (training)
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ stats, int rows, int cop, int mode, int N, int H, int W, int rows_pix,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, float momentum, float eps, int C, int Cs, float* scale, float* shift,
    float* save_mean, float* save_invstd) {
    const int c = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ double red[256];
    if (c >= C) {                      // padded channels: normalised value is exactly 0
        if (tid == 0) { scale[c] = 0.f; shift[c] = 0.f; save_mean[c] = 0.f; save_invstd[c] = 0.f; }
        return;
    }
    const int th = (mode == 0 && rows_pix > 0) ? rows_pix : 16;      // conv tile height (mode 0)
    const int tiles_x = (W + 15) / 16, tiles_y = (H + th - 1) / th;
    const long P = (long)N * H * W;
    const int planes = mode == 2 ? 3 : 2;
    auto row_count = [&](int r) -> double {
        if (mode == 2) return (double)stats[((size_t)r * 3 + 2) * cop + c];
        if (mode == 0) {
            const int tx = r % tiles_x, ty = (r / tiles_x) % tiles_y;
            const int vx = min(16, W - tx * 16), vy = min(th, H - ty * th);
            return (double)(vx * vy);
        }
        const long left = P - (long)r * rows_pix;
        return (double)(left < rows_pix ? left : rows_pix);
    };
    double s = 0.0;
    for (int r = tid; r < rows; r += 256) s += (double)stats[((size_t)r * planes) * cop + c];
    red[tid] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double mean = red[0] / (double)P;
    __syncthreads();
    double m2 = 0.0;
    for (int r = tid; r < rows; r += 256) {
        const double n = row_count(r);
        if (n <= 0.0) continue;
        const double mi = (double)stats[((size_t)r * planes) * cop + c] / n;
        m2 += (double)stats[((size_t)r * planes + 1) * cop + c] + n * (mi - mean) * (mi - mean);
    }
    red[tid] = m2; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        const double var = red[0] / (double)P;                 // biased: used for normalisation
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)((double)gamma[c] * invstd);
        scale[c] = sc;
        shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)invstd;
        if (running_mean) {
            const double unb = P > 1 ? red[0] / (double)(P - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
        }
    }
}

extern "C" int amx_bn_finalize(const float* stats, int rows, int cop, int mode, int N, int H, int W,
                               int rows_pix, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps,
                               int C, int Cs, float* scale, float* shift, float* save_mean,
                               float* save_invstd, void* stream) {
    if (!stats || !gamma || !beta || !scale || !shift || !save_mean || !save_invstd) AMX_BADARG(1);
    if (rows <= 0 || C <= 0 || Cs < C || cop < C) AMX_BADARG(2);
    if (mode < 0 || mode > 2 || (mode == 1 && rows_pix <= 0)) AMX_BADARG(3);
    AMX_LAUNCH(bn_finalize_kernel, dim3(Cs), dim3(256), 0, (hipStream_t)stream, stats, rows, cop, mode,
               N, H, W, rows_pix, gamma, beta, running_mean, running_var, momentum, eps, C, Cs, scale,
               shift, save_mean, save_invstd);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ eval-mode affine
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, int C, int Cs, float* scale,
                                      float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cs) return;
    if (c >= C) { scale[c] = 0.f; shift[c] = 0.f; return; }
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * inv;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int amx_bn_eval_affine(const float* gamma, const float* beta, const float* rm,
                                  const float* rv, float eps, int C, int Cs, float* scale, float* shift,
                                  void* stream) {
    if (!gamma || !beta || !rm || !rv || !scale || !shift || C <= 0 || Cs < C) AMX_BADARG(1);
    AMX_LAUNCH(bn_eval_affine_kernel, dim3(amx_ceil_div(Cs, 64)), dim3(64), 0, (hipStream_t)stream,
               gamma, beta, rm, rv, eps, C, Cs, scale, shift);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ y = a*scale + shift
__global__ void affine_nhwc_kernel(const float* __restrict__ a, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ y, size_t n4,
                                   int G) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % G);
        float4 v = amx_ld4(a + i * 4);
        const float4 sc = amx_ld4(scale + cg * 4), sh = amx_ld4(shift + cg * 4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        amx_st4(y + i * 4, v);
    }
}

extern "C" int amx_affine_nhwc(const float* a, const float* scale, const float* shift, float* y,
                               long npix, int Cs, void* stream) {
    if (!a || !scale || !shift || !y || (Cs & 3) || Cs <= 0) AMX_BADARG(1);
    const size_t n4 = (size_t)npix * (Cs / 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    AMX_LAUNCH(affine_nhwc_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a,
               scale, shift, y, n4, Cs / 4);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ backward: reductions
// rows: block b covers pixels [b*ppb, (b+1)*ppb); writes part[b][0][Cs] = sum dy, part[b][1][Cs] = sum dy*a
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ a, long npix,
                                                            int Cs, int ppb, float* __restrict__ part) {
    const int tid = threadIdx.x;
    PixMap m(Cs, tid);
    AMX_DYN_SMEM(float, s);                       // [2][PL][Cs]
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    if (m.active) {
        // 4 pixels (8 loads) of a thread in flight; the sums are formed in pixel order as before.  Loads are unconditional
        // (clamped to the iteration's first pixel) so that none of them is waited for where it is issued (conv1.hip).
        constexpr int U = 4;
        for (long p = p0 + m.pl; p < p1; p += (long)m.PL * U) {
            float4 g[U], v[U];
            bool ok[U];
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = p + (long)u * m.PL < p1;
                const size_t o = (size_t)(ok[u] ? p + (long)u * m.PL : p) * Cs + m.cg * 4;
                g[u] = amx_ld4(dy + o);
                v[u] = amx_ld4(a + o);
            }
            #pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                s1.x += g[u].x; s1.y += g[u].y; s1.z += g[u].z; s1.w += g[u].w;
                s2.x = fmaf(g[u].x, v[u].x, s2.x); s2.y = fmaf(g[u].y, v[u].y, s2.y);
                s2.z = fmaf(g[u].z, v[u].z, s2.z); s2.w = fmaf(g[u].w, v[u].w, s2.w);
            }
        }
    }
    if (m.active) {
        amx_st4(s + ((size_t)m.pl * Cs + m.cg * 4), s1);
        amx_st4(s + ((size_t)(m.PL + m.pl) * Cs + m.cg * 4), s2);
    }
    __syncthreads();
    for (int c = tid; c < 2 * Cs; c += 256) {
        const int which = c / Cs, ch = c - which * Cs;
        float acc = 0.f;
        for (int q = 0; q < m.PL; ++q) acc += s[(size_t)(which * m.PL + q) * Cs + ch];
        part[((size_t)blockIdx.x * 2 + which) * Cs + ch] = acc;
    }
}

static inline int pick_ppb(long npix, int* nblk) {
    int ppb = 1024;
    long nb = (npix + ppb - 1) / ppb;
    while (nb > 4096) { ppb *= 2; nb = (npix + ppb - 1) / ppb; }
    *nblk = (int)nb;
    return ppb;
}

extern "C" int amx_rows_for(long npix) { int nb; pick_ppb(npix, &nb); return nb; }
extern "C" int amx_rows_pix(long npix) { int nb; return pick_ppb(npix, &nb); }

extern "C" int amx_bn_bwd_reduce(const float* dy, const float* a, long npix, int Cs, float* part,
                                 void* stream) {
    if (!dy || !a || !part || (Cs & 3) || Cs <= 0 || Cs > 1024 || npix <= 0) AMX_BADARG(1);
    int nblk; const int ppb = pick_ppb(npix, &nblk);
    const int PL = 256 / (Cs / 4);
    AMX_LAUNCH(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), (size_t)2 * PL * Cs * sizeof(float),
               (hipStream_t)stream, dy, a, npix, Cs, ppb, part);
    AMX_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ part, int rows, int stride, int Cs, int C, double inv_n,
    const float* __restrict__ gamma, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* dgamma, float* dbeta, float* k1, float* k2, float* k3) {
    const int c = blockIdx.x, tid = threadIdx.x;
    __shared__ double r1[256], r2[256];
    if (c >= C) { if (tid == 0) { k1[c] = 0.f; k2[c] = 0.f; k3[c] = 0.f; } return; }
    double a1 = 0.0, a2 = 0.0;
    for (int r = tid; r < rows; r += 256) {
        a1 += (double)part[((size_t)r * 2) * stride + c];
        a2 += (double)part[((size_t)r * 2 + 1) * stride + c];
    }
    r1[tid] = a1; r2[tid] = a2; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double sdy = r1[0], sdya = r2[0];
        const double mean = save_mean[c], invstd = save_invstd[c], g = gamma[c];
        const double dg = invstd * (sdya - mean * sdy);
        dgamma[c] = (float)dg;
        dbeta[c] = (float)sdy;
        k1[c] = (float)(g * invstd);
        k2[c] = (float)(-g * invstd * invstd * dg * inv_n);
        k3[c] = (float)(-g * invstd * sdy * inv_n + g * invstd * invstd * mean * dg * inv_n);
    }
}

extern "C" int amx_bn_bwd_finalize(const float* part, int rows, int stride, int Cs, int C, long npix,
                                   const float* gamma, const float* save_mean,
                                   const float* save_invstd, float* dgamma, float* dbeta, float* k1,
                                   float* k2, float* k3, void* stream) {
    if (!part || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !k1 || !k2 || !k3)
        AMX_BADARG(1);
    if (stride < C) AMX_BADARG(2);
    AMX_LAUNCH(bn_bwd_finalize_kernel, dim3(Cs), dim3(256), 0, (hipStream_t)stream, part, rows, stride, Cs, C,
               1.0 / (double)npix, gamma, save_mean, save_invstd, dgamma, dbeta, k1, k2, k3);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ backward: apply
// dpre = gx + lrelu'(a) * (gx + k1*dy + k2*a + k3)      (gx == 0 unless DilatedBlock, blocks.py:321-329)
// k1/k2/k3 may be null (no BatchNorm): dpre = gx + lrelu'(a) * dy — the block then sums only (pre, a), and dy, the
// gradient of the value the consumers see (= a), already contains gx (engine.Tape.accumulate).
// part[b][Cs] = per-block sum of dpre (conv bias gradient).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ gx,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3,
    float slope, long npix, int Cs, int ppb, float* __restrict__ dpre, float* __restrict__ part) {
    const int tid = threadIdx.x;
    PixMap m(Cs, tid);
    AMX_DYN_SMEM(float, s);                       // [PL][Cs]
    float4 sb = make_float4(0, 0, 0, 0);
    const long p0 = (long)blockIdx.x * ppb;
    const long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    if (m.active) {
        float4 c1 = make_float4(1, 1, 1, 1), c2 = make_float4(0, 0, 0, 0), c3 = c2;
        if (k1) { c1 = amx_ld4(k1 + m.cg * 4); c2 = amx_ld4(k2 + m.cg * 4); c3 = amx_ld4(k3 + m.cg * 4); }
        for (long p = p0 + m.pl; p < p1; p += m.PL) {
            const size_t o = (size_t)p * Cs + m.cg * 4;
            const float4 g = amx_ld4(dy + o);
            const float4 v = amx_ld4(a + o);
            float4 e = make_float4(0, 0, 0, 0);
            if (gx) e = amx_ld4(gx + o);
            const float ei = k1 ? 1.f : 0.f;      // the activation is a summed sub-layer of its own only under a BatchNorm
            float4 d;
            d.x = e.x + (v.x > 0.f ? 1.f : slope) * (ei * e.x + fmaf(c1.x, g.x, fmaf(c2.x, v.x, c3.x)));
            d.y = e.y + (v.y > 0.f ? 1.f : slope) * (ei * e.y + fmaf(c1.y, g.y, fmaf(c2.y, v.y, c3.y)));
            d.z = e.z + (v.z > 0.f ? 1.f : slope) * (ei * e.z + fmaf(c1.z, g.z, fmaf(c2.z, v.z, c3.z)));
            d.w = e.w + (v.w > 0.f ? 1.f : slope) * (ei * e.w + fmaf(c1.w, g.w, fmaf(c2.w, v.w, c3.w)));
            amx_st4(dpre + o, d);
            sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
        }
        amx_st4(s + ((size_t)m.pl * Cs + m.cg * 4), sb);
    }
    __syncthreads();
    if (part)
        for (int c = tid; c < Cs; c += 256) {
            float acc = 0.f;
            for (int q = 0; q < m.PL; ++q) acc += s[(size_t)q * Cs + c];
            part[(size_t)blockIdx.x * Cs + c] = acc;
        }
}

extern "C" int amx_bn_bwd_apply(const float* dy, const float* a, const float* gx, const float* k1,
                                const float* k2, const float* k3, float slope, long npix, int Cs,
                                float* dpre, float* part, void* stream) {
    if (!dy || !a || !dpre || (Cs & 3) || Cs <= 0 || Cs > 1024 || npix <= 0) AMX_BADARG(1);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(2);
    int nblk; const int ppb = pick_ppb(npix, &nblk);
    const int PL = 256 / (Cs / 4);
    AMX_LAUNCH(bn_bwd_apply_kernel, dim3(nblk), dim3(256), (size_t)PL * Cs * sizeof(float),
               (hipStream_t)stream, dy, a, gx, k1, k2, k3, slope, npix, Cs, ppb, dpre, part);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ out[c] = sum_r part[r][c] (c < C)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int rows,
                                                          int stride, int C, float scale,
                                                          float* __restrict__ out) {
    const int c = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[256];
    double a = 0.0;
    for (int r = tid; r < rows; r += 256) a += (double)part[(size_t)r * stride + c];
    red[tid] = a; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) out[c] = (float)(red[0] * (double)scale);
}

extern "C" int amx_reduce_rows(const float* part, int rows, int stride, int C, float scale, float* out,
                               void* stream) {
    if (!part || !out || rows <= 0 || C <= 0 || stride < C) AMX_BADARG(1);
    AMX_LAUNCH(reduce_rows_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, rows, stride, C,
               scale, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ stage-1 merges (coalesced, parallel)
// Merges chunks of (sum, M2) rows into [RB][3][cop] rows (sum, M2 about the chunk mean, count) with Chan's
// formula in fp64, so that amx_bn_finalize (mode 2) only has to walk RB rows per channel.
// mode 3: rows of a lattice-mode convolution, [n][ry][rx][strip][tx] over the lat*lat residue-class sub-images
// (conv_kernel.h); strips of residue classes with a shorter sub-image hold no pixel and are skipped.
// Block = 16 channels x 16 row lanes: row lane j merges rows r0 + j, r0 + j + 16, ... of the chunk (64-byte row
// segments, 16 independent chains per channel), then the 16 partials are merged in lane order through LDS — fixed
// order, fp64, no atomics.  (One thread per channel walking the whole chunk, as in round 1, left 3/4 of the lanes
// idle on 16-channel layers and serialised ~128 dependent loads: 27 us per launch, 0.35 ms per training step.)
#define MRG_RL 16
__global__ __launch_bounds__(256) void bn_stats_merge_kernel(const float* __restrict__ stats, int rows,
                                                             int cop, int mode, int N, int H, int W,
                                                             int rows_pix, int lat, int chunk,
                                                             float* __restrict__ out) {
    __shared__ double sm[3][MRG_RL][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;                       // cop is a multiple of 16
    const int th = ((mode == 0 || mode == 3) && rows_pix > 0) ? rows_pix : 16;
    const int d = mode == 3 ? lat : 1;
    const int tiles_x = ((W + d - 1) / d + 15) / 16, tiles_y = ((H + d - 1) / d + th - 1) / th;
    const long P = (long)N * H * W;
    const int r0 = blockIdx.y * chunk;
    const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = r0 + rl; r < r1; r += MRG_RL) {
        double nr;
        if (mode == 0) {
            const int tx = r % tiles_x, ty = (r / tiles_x) % tiles_y;
            nr = (double)(min(16, W - tx * 16) * min(th, H - ty * th));
        } else if (mode == 3) {
            int q = r;
            const int tx = q % tiles_x; q /= tiles_x;
            const int sy = q % tiles_y; q /= tiles_y;
            const int rx = q % d, ry = (q / d) % d;
            const int Hs = (H - ry + d - 1) / d, Ws = (W - rx + d - 1) / d;
            nr = (double)(max(0, min(16, Ws - tx * 16)) * max(0, min(th, Hs - sy * th)));
            if (nr <= 0.0) continue;
        } else {
            const long left = P - (long)r * rows_pix;
            nr = (double)(left < rows_pix ? left : rows_pix);
        }
        const double mr = (double)stats[((size_t)r * 2) * cop + c] / nr;
        const double m2r = (double)stats[((size_t)r * 2 + 1) * cop + c];
        const double nt = n + nr, dl = mr - mean;
        mean += dl * (nr / nt);
        m2 += m2r + dl * dl * (n * nr / nt);
        n = nt;
    }
    sm[0][rl][cl] = n; sm[1][rl][cl] = mean; sm[2][rl][cl] = m2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < MRG_RL; ++j) {
            const double nr = sm[0][j][cl];
            if (nr <= 0.0) continue;
            const double nt = n + nr, dl = sm[1][j][cl] - mean;
            mean += dl * (nr / nt);
            m2 += sm[2][j][cl] + dl * dl * (n * nr / nt);
            n = nt;
        }
        out[((size_t)blockIdx.y * 3 + 0) * cop + c] = (float)(mean * n);
        out[((size_t)blockIdx.y * 3 + 1) * cop + c] = (float)m2;
        out[((size_t)blockIdx.y * 3 + 2) * cop + c] = (float)n;
    }
}

extern "C" int amx_bn_stats_merge(const float* stats, int rows, int cop, int mode, int N, int H, int W,
                                  int rows_pix, int lat, int nchunks, float* out, void* stream) {
    if (!stats || !out || rows <= 0 || cop <= 0 || (cop & 15) || nchunks <= 0 || (mode != 0 && mode != 1 && mode != 3)) AMX_BADARG(1);
    if (mode == 3 && (lat < 1 || rows_pix <= 0)) AMX_BADARG(2);
    const int chunk = amx_ceil_div(rows, nchunks);
    AMX_LAUNCH(bn_stats_merge_kernel, dim3(cop / 16, amx_ceil_div(rows, chunk)), dim3(256), 0,
               (hipStream_t)stream, stats, rows, cop, mode, N, H, W, rows_pix, lat, chunk, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// out[b][c] = sum over the b-th chunk of rows of part[r][c]  (thread per column: coalesced)
__global__ __launch_bounds__(256) void reduce_rows_chunked_kernel(const float* __restrict__ part, int rows,
                                                                  long ncols, int chunk,
                                                                  float* __restrict__ out) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    const int r0 = blockIdx.y * chunk;
    const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
    // 8 independent loads in flight per thread (the rows are ncols*4 bytes apart: a dependent load per iteration is
    // latency-bound at ~0.7 TB/s), summed in row order -> the result does not depend on the batching
    double a = 0.0;
    for (int r = r0; r < r1; r += 8) {
        float v[8];
        #pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r + j < r1) ? part[(size_t)(r + j) * ncols + c] : 0.f;
        #pragma unroll
        for (int j = 0; j < 8; ++j) a += (double)v[j];
    }
    out[(size_t)blockIdx.y * ncols + c] = (float)a;
}

extern "C" int amx_reduce_rows_chunked(const float* part, int rows, long ncols, int nchunks, float* out,
                                       void* stream) {
    if (!part || !out || rows <= 0 || ncols <= 0 || nchunks <= 0) AMX_BADARG(1);
    const int chunk = amx_ceil_div(rows, nchunks);
    AMX_LAUNCH(reduce_rows_chunked_kernel, dim3((unsigned)((ncols + 255) / 256), amx_ceil_div(rows, chunk)),
               dim3(256), 0, (hipStream_t)stream, part, rows, ncols, chunk, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ column sums of SEVERAL partial-row tensors at once
// The rVAE decoder backward leaves seven per-sample partial tensors part_k[rows][cols_k] (dW, db per hidden layer, dWo,
// dbo, dWc, dbc, dWz; rdecoder.hip); summing each with its own pair of launches put 14 five-to-eighteen-microsecond kernels on the
// stream after every backward (profiles/r03_rvae_step_timeline.txt).  Here the column spaces are concatenated: ONE
// stage-1 launch (chunk sums in row order, fp64 accumulation) and ONE stage-2 launch (chunks in order) serve all of
// them; out_k receives cols_k floats.  Same arithmetic and order as amx_reduce_rows_chunked twice -> identical values.
#define AMX_MAXSEG 16
struct SegArgs { const float* part[AMX_MAXSEG]; float* out[AMX_MAXSEG]; long stride[AMX_MAXSEG]; long start[AMX_MAXSEG + 1]; int nseg; };

__global__ __launch_bounds__(256) void reduce_rows_segments_kernel(SegArgs s, int rows, int chunk, float* __restrict__ tmp) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= s.start[s.nseg]) return;
    int k = 0;
    while (j >= s.start[k + 1]) ++k;
    const long c = j - s.start[k], ncols = s.stride[k];          // (row stride of this segment, >= its column count)
    const float* part = s.part[k];
    const int r0 = blockIdx.y * chunk;
    const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
    double a = 0.0;
    for (int r = r0; r < r1; r += 8) {
        float v[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (r + q < r1) ? part[(size_t)(r + q) * ncols + c] : 0.f;
        #pragma unroll
        for (int q = 0; q < 8; ++q) a += (double)v[q];
    }
    tmp[(size_t)blockIdx.y * s.start[s.nseg] + j] = (float)a;
}

__global__ __launch_bounds__(256) void reduce_rows_segments_final_kernel(SegArgs s, int nch, const float* __restrict__ tmp) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = s.start[s.nseg];
    if (j >= tot) return;
    int k = 0;
    while (j >= s.start[k + 1]) ++k;
    double a = 0.0;
    for (int r = 0; r < nch; ++r) a += (double)tmp[(size_t)r * tot + j];
    s.out[k][j - s.start[k]] = (float)a;
}

// tmp: nchunks * (sum of cols) floats; strides[k] >= cols[k]: floats between consecutive rows of parts[k]
extern "C" int amx_reduce_rows_segments(const float* const* parts, const long* cols, const long* strides,
                                        float* const* outs, int nseg, int rows, int nchunks, float* tmp, void* stream) {
    if (!parts || !cols || !strides || !outs || !tmp || nseg < 1 || nseg > AMX_MAXSEG || rows <= 0 || nchunks <= 0) AMX_BADARG(1);
    SegArgs s;
    s.nseg = nseg; s.start[0] = 0;
    for (int k = 0; k < AMX_MAXSEG; ++k) { s.part[k] = nullptr; s.out[k] = nullptr; s.stride[k] = 0; }
    for (int k = 0; k < nseg; ++k) {
        if (!parts[k] || !outs[k] || cols[k] <= 0 || strides[k] < cols[k]) AMX_BADARG(2);
        s.part[k] = parts[k]; s.out[k] = outs[k]; s.stride[k] = strides[k]; s.start[k + 1] = s.start[k] + cols[k];
    }
    const int chunk = amx_ceil_div(rows, nchunks);
    const int nch = amx_ceil_div(rows, chunk);
    const unsigned nb = (unsigned)((s.start[nseg] + 255) / 256);
    AMX_LAUNCH(reduce_rows_segments_kernel, dim3(nb, nch), dim3(256), 0, (hipStream_t)stream, s, rows, chunk, tmp);
    AMX_CHECK_LAUNCH();
    AMX_LAUNCH(reduce_rows_segments_final_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, s, nch, tmp);
    AMX_CHECK_LAUNCH();
    return 0;
}
