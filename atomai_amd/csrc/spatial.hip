// spatial.hip — max-pool, x2 upsampling (bilinear / nearest) and the DilatedBlock sum, NHWC float4.
// All HBM-bound elementwise/gather kernels: one float4 (4 channels) per thread, consecutive threads
// on consecutive channel groups -> fully coalesced 16 B/lane accesses.
//
//   F.max_pool2d(x, 2, 2) on the BN output                                  atomai/nets/fcnn.py:123-127, 219
//   F.interpolate(scale_factor=2, mode=bilinear|nearest), align_corners=False   atomai/nets/blocks.py:130-131
//   DilatedBlock: sum of every sub-layer output                              atomai/nets/blocks.py:321-329
#include "amx_device.h"
#include <cstdlib>

#define GRID_FOR(n) dim3((unsigned)(((n) + 255) / 256 < 16384 ? ((n) + 255) / 256 : 16384))

// ------------------------------------------------------------------ max pool 2x2 / stride 2
// Reads the RAW post-activation tensor a plus the producer's BN affine (scale may be negative, so the
// affine is applied before the max); writes the normalised pooled tensor.
__global__ void pool_fwd_kernel(const float* __restrict__ a, const float* __restrict__ scale,
                                const float* __restrict__ shift, float* __restrict__ d, int N, int H,
                                int W, int G) {
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % G);
        size_t r = i / G;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho); const int n = (int)(r / Ho);
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (scale) { sc = amx_ld4(scale + cg * 4); sh = amx_ld4(shift + cg * 4); }
        float4 best;
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = 2 * yo + (k >> 1), x = 2 * xo + (k & 1);
            float4 v = amx_ld4(a + (((size_t)n * H + y) * W + x) * G * 4 + cg * 4);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (k == 0) best = v;
            else {
                best.x = v.x > best.x ? v.x : best.x; best.y = v.y > best.y ? v.y : best.y;
                best.z = v.z > best.z ? v.z : best.z; best.w = v.w > best.w ? v.w : best.w;
            }
        }
        amx_st4(d + i * 4, best);
    }
}

extern "C" int amx_pool2x2_fwd(const float* a, const float* scale, const float* shift, float* d, int N,
                               int H, int W, int Cs, void* stream) {
    if (!a || !d || (Cs & 3) || Cs <= 0 || H < 2 || W < 2) AMX_BADARG(1);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(2);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (Cs / 4);
    AMX_LAUNCH(pool_fwd_kernel, GRID_FOR(total), dim3(256), 0, (hipStream_t)stream, a, scale, shift, d,
               N, H, W, Cs / 4);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Backward: dy[full res] = skip_grad (optional) + route(g) where the gradient of each 2x2 window goes
// to its FIRST maximum in scan order (torch semantics); the arg-max is recomputed from a + affine.
// bstats (optional): [gridDim.x][2][4G] per-block (sum dy, sum dy*a) of the producer layer's BatchNorm backward;
// requires (gridDim.x * 256) % G == 0 so that a thread keeps its channel group.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                const float* __restrict__ scale, const float* __restrict__ shift,
                                const float* __restrict__ skip, float* __restrict__ dy, int N, int H,
                                int W, int G, float* __restrict__ bstats) {
    float4 bs1 = make_float4(0, 0, 0, 0), bs2 = make_float4(0, 0, 0, 0);
    const int Ho = H >> 1, Wo = W >> 1;
    const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;      // also cover an odd last row/col (skip only)
    const size_t total = (size_t)N * Hc * Wc * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % G);
        size_t r = i / G;
        const int xo = (int)(r % Wc); r /= Wc;
        const int yo = (int)(r % Hc); const int n = (int)(r / Hc);
        const bool inwin = (yo < Ho) && (xo < Wo);
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (scale) { sc = amx_ld4(scale + cg * 4); sh = amx_ld4(shift + cg * 4); }
        float4 v[4], raw[4];
        bool ok[4];
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = 2 * yo + (k >> 1), x = 2 * xo + (k & 1);
            ok[k] = (y < H) && (x < W);
            v[k] = make_float4(0, 0, 0, 0);
            raw[k] = v[k];
            if (ok[k] && (inwin || bstats)) {
                float4 t = amx_ld4(a + (((size_t)n * H + y) * W + x) * G * 4 + cg * 4);
                raw[k] = t;
                t.x = fmaf(t.x, sc.x, sh.x); t.y = fmaf(t.y, sc.y, sh.y);
                t.z = fmaf(t.z, sc.z, sh.z); t.w = fmaf(t.w, sc.w, sh.w);
                v[k] = t;
            }
        }
        int ax = 0, ay = 0, az = 0, aw = 0;
        float4 gg = make_float4(0, 0, 0, 0);
        if (inwin) {
            float4 best = v[0];
            #pragma unroll
            for (int k = 1; k < 4; ++k) {
                if (v[k].x > best.x) { best.x = v[k].x; ax = k; }
                if (v[k].y > best.y) { best.y = v[k].y; ay = k; }
                if (v[k].z > best.z) { best.z = v[k].z; az = k; }
                if (v[k].w > best.w) { best.w = v[k].w; aw = k; }
            }
            gg = amx_ld4(g + (((size_t)n * Ho + yo) * Wo + xo) * G * 4 + cg * 4);
        }
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const int y = 2 * yo + (k >> 1), x = 2 * xo + (k & 1);
            const size_t o = (((size_t)n * H + y) * W + x) * G * 4 + cg * 4;
            float4 out = make_float4(0, 0, 0, 0);
            if (skip) out = amx_ld4(skip + o);
            if (inwin) {
                out.x += ax == k ? gg.x : 0.f; out.y += ay == k ? gg.y : 0.f;
                out.z += az == k ? gg.z : 0.f; out.w += aw == k ? gg.w : 0.f;
            }
            amx_st4(dy + o, out);
            bs1.x += out.x; bs1.y += out.y; bs1.z += out.z; bs1.w += out.w;
            bs2.x = fmaf(out.x, raw[k].x, bs2.x); bs2.y = fmaf(out.y, raw[k].y, bs2.y);
            bs2.z = fmaf(out.z, raw[k].z, bs2.z); bs2.w = fmaf(out.w, raw[k].w, bs2.w);
        }
    }
    if (!bstats) return;
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x;
    red[0][tid] = bs1; red[1][tid] = bs2;
    __syncthreads();
    if (tid < 2 * G) {                           // thread (which, cg): fixed-order sum over the block's threads of cg
        const int which = tid / G, cgq = tid - which * G;
        float4 t = make_float4(0, 0, 0, 0);
        for (int q = cgq; q < 256; q += G) { const float4 u = red[which][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        amx_st4(bstats + ((size_t)blockIdx.x * 2 + which) * (G * 4) + cgq * 4, t);
    }
}

static int pool_bwd_blocks(size_t total) {
    size_t nb = (total + 255) / 256;
    return (int)(nb < 2048 ? nb : 2048);
}

extern "C" int amx_pool2x2_bwd_rows(int N, int H, int W, int Cs) {
    return pool_bwd_blocks((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (Cs / 4));
}

extern "C" int amx_pool2x2_bwd(const float* g, const float* a, const float* scale, const float* shift,
                               const float* skip, float* dy, float* bstats, int N, int H, int W, int Cs,
                               void* stream) {
    if (!g || !a || !dy || (Cs & 3) || Cs <= 0 || H < 2 || W < 2) AMX_BADARG(1);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(2);
    if (bstats && (256 % (Cs / 4)) != 0) AMX_BADARG(3);
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (Cs / 4);
    AMX_LAUNCH(pool_bwd_kernel, dim3(pool_bwd_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, a, scale,
               shift, skip, dy, N, H, W, Cs / 4, bstats);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ pool backward + the FIRST layer's weight gradient
// U-Net / dilnet / SegResNet start with conv(1 -> F) -> LeakyReLU -> BatchNorm -> max-pool; the net input needs no
// gradient, so the pooling backward's output dy (the complete gradient of that layer's output) has exactly ONE reader: the
// first-layer weight-gradient kernel, which reads it back together with the activation at the very END of the backward pass,
// where no other work is left to overlap it (pool_bwd 0.40 + conv1_wgrad 0.33 ms of a 16.7 ms U-Net step,
// profiles/r06_step_timeline_final.txt).  The weight gradient is LINEAR in the three per-channel constants of the BatchNorm
// backward, which are only known after this pass (they need the batch sums it produces):
//     dW[c][t] = sum_p lrelu'(a) (k1 dy + k2 a + k3) x[p + t] = k1[c] S1[c][t] + k2[c] S2[c][t] + k3[c] S3[c][t],
//     S1 = sum lrelu'(a) dy x_t,   S2 = sum lrelu'(a) a x_t,   S3 = sum lrelu'(a) x_t       (t = 9: x_t := 1, the bias row)
// so this kernel forms dy in registers as pool_bwd_kernel does, adds up S1..S3 next to the BatchNorm-backward sums, and
// never writes dy: one pass over (a, skip gradient, pooled gradient, x) instead of two passes and a 4 * Cs B/pixel tensor.
// part3 [gridDim.x][3][10][4G]; bstats [gridDim.x][2][4G]; even H and W, dilation 1, G a power of two <= 16.
__global__ __launch_bounds__(256) void pool_bwd_wgrad1_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ skip, const float* __restrict__ x,
                                                              float slope, int N, int H, int W, int G,
                                                              float* __restrict__ bstats, float* __restrict__ part3) {
    float4 acc[32];                                   // 0, 1: sum dy, sum dy * a;  2 + 10 j + t: S_{j+1}[t]
    #pragma unroll
    for (int r = 0; r < 32; ++r) acc[r] = make_float4(0, 0, 0, 0);
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % G);
        size_t r = i / G;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho); const int n = (int)(r / Ho);
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (scale) { sc = amx_ld4(scale + cg * 4); sh = amx_ld4(shift + cg * 4); }
        float4 raw[4], v[4], sk[4];
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = (((size_t)n * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * G * 4 + cg * 4;
            raw[k] = amx_ld4(a + o);
            sk[k] = skip ? amx_ld4(skip + o) : make_float4(0, 0, 0, 0);
        }
        const float4 gg = amx_ld4(g + (((size_t)n * Ho + yo) * Wo + xo) * G * 4 + cg * 4);
        float xp[4][4];                               // input patch rows 2yo-1 .. 2yo+2, columns 2xo-1 .. 2xo+2 (zero padding)
        #pragma unroll
        for (int py = 0; py < 4; ++py)
            #pragma unroll
            for (int px = 0; px < 4; ++px) {
                const int yy = 2 * yo - 1 + py, xx = 2 * xo - 1 + px;
                const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
                xp[py][px] = in ? x[((size_t)n * H + yy) * W + xx] : 0.f;
            }
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 t = raw[k];
            t.x = fmaf(t.x, sc.x, sh.x); t.y = fmaf(t.y, sc.y, sh.y);
            t.z = fmaf(t.z, sc.z, sh.z); t.w = fmaf(t.w, sc.w, sh.w);
            v[k] = t;
        }
        int ax = 0, ay = 0, az = 0, aw = 0;           // first maximum in scan order (torch semantics), as pool_bwd_kernel
        float4 best = v[0];
        #pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (v[k].x > best.x) { best.x = v[k].x; ax = k; }
            if (v[k].y > best.y) { best.y = v[k].y; ay = k; }
            if (v[k].z > best.z) { best.z = v[k].z; az = k; }
            if (v[k].w > best.w) { best.w = v[k].w; aw = k; }
        }
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 dy = sk[k];
            dy.x += ax == k ? gg.x : 0.f; dy.y += ay == k ? gg.y : 0.f;
            dy.z += az == k ? gg.z : 0.f; dy.w += aw == k ? gg.w : 0.f;
            const float4 rw = raw[k];
            acc[0].x += dy.x; acc[0].y += dy.y; acc[0].z += dy.z; acc[0].w += dy.w;
            acc[1].x = fmaf(dy.x, rw.x, acc[1].x); acc[1].y = fmaf(dy.y, rw.y, acc[1].y);
            acc[1].z = fmaf(dy.z, rw.z, acc[1].z); acc[1].w = fmaf(dy.w, rw.w, acc[1].w);
            float4 u[3];
            u[2] = make_float4(rw.x > 0.f ? 1.f : slope, rw.y > 0.f ? 1.f : slope, rw.z > 0.f ? 1.f : slope, rw.w > 0.f ? 1.f : slope);
            u[0] = make_float4(u[2].x * dy.x, u[2].y * dy.y, u[2].z * dy.z, u[2].w * dy.w);
            u[1] = make_float4(u[2].x * rw.x, u[2].y * rw.y, u[2].z * rw.z, u[2].w * rw.w);
            #pragma unroll
            for (int j = 0; j < 3; ++j) {
                float4& b = acc[2 + 10 * j + 9];
                b.x += u[j].x; b.y += u[j].y; b.z += u[j].z; b.w += u[j].w;
                #pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float xv = xp[(k >> 1) + t / 3][(k & 1) + t % 3];
                    float4& s4 = acc[2 + 10 * j + t];
                    s4.x = fmaf(xv, u[j].x, s4.x); s4.y = fmaf(xv, u[j].y, s4.y);
                    s4.z = fmaf(xv, u[j].z, s4.z); s4.w = fmaf(xv, u[j].w, s4.w);
                }
            }
        }
    }
    // the 64 / G lanes of a wave that share a channel group (a thread keeps its group: (gridDim.x * 256) % G == 0), then
    // the four waves through LDS in wave order: fixed order, no atomics
    __shared__ float4 red[4][32][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    #pragma unroll
    for (int r = 0; r < 32; ++r) {
        float4 t = acc[r];
        for (int o = G; o < 64; o <<= 1) {
            t.x += __shfl_xor(t.x, o); t.y += __shfl_xor(t.y, o); t.z += __shfl_xor(t.z, o); t.w += __shfl_xor(t.w, o);
        }
        if (lane < G) red[wave][r][lane] = t;
    }
    __syncthreads();
    for (int e = tid; e < 32 * G; e += 256) {
        const int r = e / G, cgq = e - r * G;
        float4 t = red[0][r][cgq];
        #pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 q = red[w][r][cgq]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
        if (r < 2) amx_st4(bstats + ((size_t)blockIdx.x * 2 + r) * (G * 4) + cgq * 4, t);
        else amx_st4(part3 + ((size_t)blockIdx.x * 30 + (r - 2)) * (G * 4) + cgq * 4, t);
    }
}

extern "C" int amx_pool2x2_bwd_wgrad1_supported(int H, int W, int Cs, int dil) {
    const int G = Cs >> 2;
    return (Cs > 0 && !(Cs & 3) && G <= 16 && (G & (G - 1)) == 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && dil == 1) ? 1 : 0;
}

// rows of bstats / part3 = amx_pool2x2_bwd_rows(N, H, W, Cs)
extern "C" int amx_pool2x2_bwd_wgrad1(const float* g, const float* a, const float* scale, const float* shift,
                                      const float* skip, const float* x, float slope, float* bstats, float* part3, int N,
                                      int H, int W, int Cs, void* stream) {
    if (!g || !a || !x || !bstats || !part3) AMX_BADARG(1);
    if (!amx_pool2x2_bwd_wgrad1_supported(H, W, Cs, 1)) AMX_BADARG(2);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (Cs / 4);
    AMX_LAUNCH(pool_bwd_wgrad1_kernel, dim3(pool_bwd_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, a, scale, shift,
               skip, x, slope, N, H, W, Cs / 4, bstats, part3);
    AMX_CHECK_LAUNCH();
    return 0;
}

// out [10][Cs] = k1 S1 + k2 S2 + k3 S3 (fp64, rows in order) from (chunk-reduced) part3 rows [rows][3][10][Cs];
// k1 == NULL (no BatchNorm): out = S1.  The row layout of amx_conv1_wgrad_fused's column sums: taps 0..8, then the bias row.
__global__ __launch_bounds__(256) void conv1_wgrad_combine_kernel(const float* __restrict__ part3, int rows, int Cs,
                                                                  const float* __restrict__ k1, const float* __restrict__ k2,
                                                                  const float* __restrict__ k3, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 10 * Cs) return;
    const int c = e % Cs;
    double s[3] = {0.0, 0.0, 0.0};
    for (int r = 0; r < rows; ++r)
        #pragma unroll
        for (int j = 0; j < 3; ++j) s[j] += (double)part3[((size_t)r * 3 + j) * 10 * Cs + e];
    out[e] = k1 ? (float)((double)k1[c] * s[0] + (double)k2[c] * s[1] + (double)k3[c] * s[2]) : (float)s[0];
}

extern "C" int amx_conv1_wgrad_combine(const float* part3, int rows, int Cs, const float* k1, const float* k2,
                                       const float* k3, float* out, void* stream) {
    if (!part3 || !out || rows <= 0 || Cs <= 0 || (Cs & 3)) AMX_BADARG(1);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(2);
    AMX_LAUNCH(conv1_wgrad_combine_kernel, dim3(amx_ceil_div(10 * Cs, 256)), dim3(256), 0, (hipStream_t)stream, part3, rows, Cs,
               k1, k2, k3, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ upsample x2
// bilinear, align_corners=False, scale 2:  out(2i)   = .25*in(i-1) + .75*in(i)
//                                          out(2i+1) = .75*in(i)   + .25*in(i+1)   (indices clamped)
__device__ __forceinline__ void up_taps(int o, int n, int mode, int& i0, int& i1, float& w0, float& w1) {
    if (mode == 1) { i0 = i1 = o >> 1; w0 = 1.f; w1 = 0.f; return; }        // nearest: floor(o/2)
    const int i = o >> 1;
    if (o & 1) { i0 = i; i1 = i + 1 < n ? i + 1 : n - 1; w0 = 0.75f; w1 = 0.25f; }
    else { i0 = i > 0 ? i - 1 : 0; i1 = i; w0 = 0.25f; w1 = 0.75f; }
}

// One workgroup = 256 (x, channel group) slots of UP_LR low-res rows = 2 * UP_LR output rows (blockIdx.x = n * ceil(h / UP_LR)
// + row group, blockIdx.y = slot chunk): a thread keeps its two low-res columns of the UP_LR + 2 (clamped) low-res rows in
// registers and writes 2 * UP_LR outputs.  The only integer divisions are two 32-bit ones per thread (the flat 1-D mapping
// of round 1 ran three 64-bit div / mod chains per 16-byte store: 3.8 TB/s).  Round 3: one output row per workgroup meant
// 4 loads and ONE 16-byte store per thread and 0.9 M workgroups per 16 frames of 1024^2 — dispatch-bound at 3.2-3.8 TB/s;
// with two low-res rows per thread: 8 loads for 4 stores and a quarter of the workgroups.
#ifndef UP_LR
#define UP_LR 2
#endif
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ v, float* __restrict__ u, int N,
                                                           int h, int w, int G, int mode, int groups) {
    const int W = 2 * w;
    const unsigned slot = blockIdx.y * 256u + threadIdx.x;       // x * G + cg within the row
    if (slot >= (unsigned)(W * G)) return;
    const int x = (int)(slot / (unsigned)G), cg = (int)(slot - (unsigned)x * G);
    const int n = (int)(blockIdx.x / (unsigned)groups), k0 = (int)(blockIdx.x - (unsigned)n * groups) * UP_LR;
    int x0, x1; float wx0, wx1;
    up_taps(x, w, mode, x0, x1, wx0, wx1);
    const float* base = v + (size_t)n * h * w * G * 4 + cg * 4;
    // low-res rows k0 - 1 .. k0 + UP_LR, clamped into the image: out(2k) = .25 A[k-1] + .75 A[k], out(2k+1) = .75 A[k] +
    // .25 A[k+1] — the (i0, i1, w0, w1) up_taps returns for these two output rows, with the same clamping
    float4 A0[UP_LR + 2], A1[UP_LR + 2];
    #pragma unroll
    for (int r = 0; r < UP_LR + 2; ++r) {
        int row = k0 - 1 + r;
        row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
        A0[r] = amx_ld4(base + ((size_t)row * w + x0) * G * 4);
        A1[r] = mode == 1 ? A0[r] : amx_ld4(base + ((size_t)row * w + x1) * G * 4);
    }
    #pragma unroll
    for (int j = 0; j < UP_LR; ++j) {
        const int k = k0 + j;
        if (k >= h) break;
        #pragma unroll
        for (int odd = 0; odd < 2; ++odd) {
            const float4 a00 = A0[j + odd], a01 = A1[j + odd], a10 = A0[j + odd + 1], a11 = A1[j + odd + 1];
            const float wy0 = odd ? 0.75f : 0.25f, wy1 = odd ? 0.25f : 0.75f;
            float4 o;
            if (mode == 1) { o = A0[j + 1]; }                 // nearest: floor(y / 2) = k
            else {
                // same association order as ATen's upsample_bilinear2d: rows first, then columns
                o.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
                o.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
                o.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
                o.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
            }
            amx_st4(u + (((size_t)n * 2 * h + 2 * k + odd) * W * G + slot) * 4, o);
        }
    }
}

extern "C" int amx_upsample2x_fwd(const float* v, float* u, int N, int h, int w, int Cs, int mode,
                                  void* stream) {
    if (!v || !u || (Cs & 3) || Cs <= 0 || h <= 0 || w <= 0 || (mode != 0 && mode != 1)) AMX_BADARG(1);
    const int G = Cs / 4;
    if ((long)N * 2 * h >= 2147483647L || (long)2 * w * G >= 2147483647L) AMX_BADARG(2);
    if (amx_ceil_div(2 * w * G, 256) > 65535) AMX_BADARG(3);
    const int groups = amx_ceil_div(h, UP_LR);
    AMX_LAUNCH(upsample_fwd_kernel, dim3((unsigned)(N * groups), amx_ceil_div(2 * w * G, 256)), dim3(256), 0,
               (hipStream_t)stream, v, u, N, h, w, G, mode, groups);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Backward as a deterministic GATHER (ATen's CUDA backward scatters with atomics): each low-res pixel
// collects from the <= 4x4 high-res pixels it contributed to.
__device__ __forceinline__ int up_bwd_taps(int i, int n, int mode, int* o, float* wgt) {
    if (mode == 1) { o[0] = 2 * i; wgt[0] = 1.f; o[1] = 2 * i + 1; wgt[1] = 1.f; return 2; }
    int c = 0;
    // out(2i) gets .75 in(i); out(2i+1) gets .75 in(i)
    o[c] = 2 * i; wgt[c++] = 0.75f + (i == 0 ? 0.25f : 0.f);            // clamp at the low edge
    o[c] = 2 * i + 1; wgt[c++] = 0.75f + (i == n - 1 ? 0.25f : 0.f);    // clamp at the high edge
    if (i > 0) { o[c] = 2 * i - 1; wgt[c++] = 0.25f; }                  // out(2(i-1)+1) uses in(i) * .25
    if (i + 1 < n) { o[c] = 2 * i + 2; wgt[c++] = 0.25f; }              // out(2(i+1))   uses in(i) * .25
    return c;
}

// One workgroup owns a 2-D patch of TY x TX low-res pixels (all channel groups), so the high-res rows that vertically
// adjacent outputs share are fetched once per workgroup and re-used from the CU's L1 (a row-major 1-D mapping re-reads
// every high-res row from L2/HBM for the rows above and below: 2x over-fetch measured with the PMC counters).
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ du, float* __restrict__ dv, int N,
                                                           int h, int w, int G, int mode, int TX, int TY,
                                                           int tiles_x, int tiles_y) {
    const int H = 2 * h, W = 2 * w;
    const int tid = threadIdx.x;
    const int cg = tid % G, lx = (tid / G) % TX, ly = tid / (G * TX);
    if (ly >= TY) return;
    int t = blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y; const int n = t / tiles_y;
    const int x = bx * TX + lx, y = by * TY + ly;
    if (x >= w || y >= h) return;
    int oy[4], ox[4]; float wy[4], wx[4];
    const int ny = up_bwd_taps(y, h, mode, oy, wy);
    const int nx = up_bwd_taps(x, w, mode, ox, wx);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            const float4 g = amx_ld4(du + (((size_t)n * H + oy[a]) * W + ox[b]) * G * 4 + cg * 4);
            const float wt = wy[a] * wx[b];
            acc.x = fmaf(wt, g.x, acc.x); acc.y = fmaf(wt, g.y, acc.y);
            acc.z = fmaf(wt, g.z, acc.z); acc.w = fmaf(wt, g.w, acc.w);
        }
    amx_st4(dv + ((((size_t)n * h + y) * w + x) * G + cg) * 4, acc);
}

extern "C" int amx_upsample2x_bwd(const float* du, float* dv, int N, int h, int w, int Cs, int mode,
                                  void* stream) {
    if (!du || !dv || (Cs & 3) || Cs <= 0 || Cs > 1024 || h <= 0 || w <= 0 || (mode != 0 && mode != 1)) AMX_BADARG(1);
    const int G = Cs / 4;
    int TX = 32 / G; if (TX < 1) TX = 1;
    int TY = 256 / (G * TX); if (TY > 8) TY = 8;
    const int tiles_x = amx_ceil_div(w, TX), tiles_y = amx_ceil_div(h, TY);
    const size_t blocks = (size_t)N * tiles_x * tiles_y;
    if (blocks >= 2147483647ull) AMX_BADARG(2);
    AMX_LAUNCH(upsample_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, du, dv, N, h, w, G, mode,
               TX, TY, tiles_x, tiles_y);
    AMX_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------ DilatedBlock sum
// out (+)= sum_i [ pre_i + a_i + bn_i ],  pre_i = a_i > 0 ? a_i : a_i / slope  (inverse LeakyReLU),
// bn_i = a_i*scale_i + shift_i (omitted when the block has no BatchNorm: scale_i == nullptr).
// amx_dilated_sum_ex: out (+)= sum_i [ wpre * pre_i + wact * a_i + bn_i ] — a Dropout layer inside the block is one
// more sub-layer whose output is summed (eval mode: Dropout is the identity, so pre_i counts twice: wpre = 2; training:
// the un-dropped convolution output is one extra pre term: a second call with wpre = 1, wact = 0 on the unmasked tensors).
struct DilSumArgs {
    const float* a[4]; const float* scale[4]; const float* shift[4];
    int n; int accumulate; float inv_slope; float wpre; float wact;
};

__global__ void dilated_sum_kernel(DilSumArgs p, float* __restrict__ out, size_t n4, int G) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % G);
        float4 acc = make_float4(0, 0, 0, 0);
        if (p.accumulate) acc = amx_ld4(out + i * 4);
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= p.n) break;
            const float4 v = amx_ld4(p.a[k] + i * 4);
            float4 sc = make_float4(0, 0, 0, 0), sh = sc;
            if (p.scale[k]) { sc = amx_ld4(p.scale[k] + cg * 4); sh = amx_ld4(p.shift[k] + cg * 4); }
            if (p.wpre == 1.f && p.wact == 1.f) {            // the plain block: the original expression, bit for bit
                acc.x += (v.x > 0.f ? v.x : v.x * p.inv_slope) + v.x + fmaf(v.x, sc.x, sh.x);
                acc.y += (v.y > 0.f ? v.y : v.y * p.inv_slope) + v.y + fmaf(v.y, sc.y, sh.y);
                acc.z += (v.z > 0.f ? v.z : v.z * p.inv_slope) + v.z + fmaf(v.z, sc.z, sh.z);
                acc.w += (v.w > 0.f ? v.w : v.w * p.inv_slope) + v.w + fmaf(v.w, sc.w, sh.w);
            } else {
                acc.x += p.wpre * (v.x > 0.f ? v.x : v.x * p.inv_slope) + p.wact * v.x + fmaf(v.x, sc.x, sh.x);
                acc.y += p.wpre * (v.y > 0.f ? v.y : v.y * p.inv_slope) + p.wact * v.y + fmaf(v.y, sc.y, sh.y);
                acc.z += p.wpre * (v.z > 0.f ? v.z : v.z * p.inv_slope) + p.wact * v.z + fmaf(v.z, sc.z, sh.z);
                acc.w += p.wpre * (v.w > 0.f ? v.w : v.w * p.inv_slope) + p.wact * v.w + fmaf(v.w, sc.w, sh.w);
            }
        }
        amx_st4(out + i * 4, acc);
    }
}

extern "C" int amx_dilated_sum_ex(const float* const* a, const float* const* scale,
                                  const float* const* shift, int n, float slope, int accumulate, float wpre,
                                  float wact, float* out, long npix, int Cs, void* stream);

extern "C" int amx_dilated_sum(const float* const* a, const float* const* scale,
                               const float* const* shift, int n, float slope, int accumulate,
                               float* out, long npix, int Cs, void* stream) {
    return amx_dilated_sum_ex(a, scale, shift, n, slope, accumulate, 1.f, 1.f, out, npix, Cs, stream);
}

extern "C" int amx_dilated_sum_ex(const float* const* a, const float* const* scale,
                                  const float* const* shift, int n, float slope, int accumulate, float wpre,
                                  float wact, float* out, long npix, int Cs, void* stream) {
    if (!a || !out || n < 1 || n > 4 || (Cs & 3) || Cs <= 0 || slope == 0.f) AMX_BADARG(1);
    DilSumArgs p;
    for (int k = 0; k < 4; ++k) {
        p.a[k] = k < n ? a[k] : nullptr;
        p.scale[k] = (k < n && scale) ? scale[k] : nullptr;
        p.shift[k] = (k < n && shift) ? shift[k] : nullptr;
    }
    p.n = n; p.accumulate = accumulate; p.inv_slope = 1.0f / slope; p.wpre = wpre; p.wact = wact;
    const size_t n4 = (size_t)npix * (Cs / 4);
    AMX_LAUNCH(dilated_sum_kernel, GRID_FOR(n4), dim3(256), 0, (hipStream_t)stream, p, out, n4, Cs / 4);
    AMX_CHECK_LAUNCH();
    return 0;
}
