// dropout.hip — training-mode nn.Dropout of ConvBlock (Conv2d -> Dropout(p) -> LeakyReLU -> BatchNorm2d,
// atomai/nets/blocks.py:59-76; Unet(dropout=True) uses it in c3 / bn / c4, fcnn.py:76-101).
//
// LeakyReLU is positively homogeneous, so lrelu(m * c) == m * lrelu(c) for the dropout multiplier m in {0, 1/(1-p)}:
// the convolution kernel runs unchanged (bias + activation fused) and ONE elementwise pass afterwards applies the mask
// in place, keeps it for the backward pass and produces the BatchNorm statistics of the masked tensor (rows in the
// "mode 1" layout of bn.hip: one (sum, M2) row per `rows_pix` consecutive pixels).  Backward multiplies the
// pre-activation gradient by the same mask.  Per-element randomness: Philox4x32-10 keyed by (seed, element index / 4),
// or an injected mask (tests).  Non-default path of the reference (dropout=False), so it is built for correctness and
// HBM efficiency, not fused into the MFMA kernels.
#include "amx_device.h"

static __device__ __forceinline__ unsigned dr_mulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static __device__ __forceinline__ void dr_philox(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned out[4]) {
    unsigned c2 = 0x64726f70u, c3 = 0u;                           // stream tag "drop"
    #pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned h0 = dr_mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = dr_mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one workgroup per statistics row (rows_pix pixels x Cs channels); G = Cs / 4 channel groups
__global__ __launch_bounds__(256) void dropout_fwd_kernel(float* __restrict__ a, float* __restrict__ mask,
                                                          const float* __restrict__ mask_in, float p, float scale,
                                                          unsigned k0, unsigned k1, float* __restrict__ stats,
                                                          long npix, int Cs, int cop, int rows_pix) {
    __shared__ float4 red[256];
    const int G = Cs >> 2;
    const int P = 256 / G;                                        // pixels handled per sweep (threads >= P*G idle)
    const int tid = threadIdx.x;
    const int cg = tid % G, lp = tid / G;
    const long pix0 = (long)blockIdx.x * rows_pix;
    const long pixn = pix0 + rows_pix < npix ? pix0 + rows_pix : npix;
    const bool act = lp < P;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        for (long px = pix0 + lp; px < pixn; px += P) {
            const long e4 = px * G + cg;                          // float4 index
            float4 v = amx_ld4(a + e4 * 4), m;
            if (mask_in) m = amx_ld4(mask_in + e4 * 4);
            else {
                unsigned r[4];
                dr_philox(k0, k1, (unsigned)(e4 & 0xffffffffL), (unsigned)(e4 >> 32), r);
                const float thr = p * 4294967296.0f;
                m.x = (float)r[0] >= thr ? scale : 0.f; m.y = (float)r[1] >= thr ? scale : 0.f;
                m.z = (float)r[2] >= thr ? scale : 0.f; m.w = (float)r[3] >= thr ? scale : 0.f;
            }
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            amx_st4(a + e4 * 4, v);
            amx_st4(mask + e4 * 4, m);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (!stats) return;
    // per-channel sums of the row: fixed-order reduction over the P threads that share a channel group
    red[tid] = s;
    __syncthreads();
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < P; ++j) { const float4 t = red[j * G + cg]; tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w; }
    __syncthreads();
    const float inv = 1.0f / (float)(pixn - pix0);
    const float4 mu = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        for (long px = pix0 + lp; px < pixn; px += P) {
            const float4 v = amx_ld4(a + (px * G + cg) * 4);      // just written: L2-resident
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
    }
    red[tid] = q;
    __syncthreads();
    if (tid < G) {
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < P; ++j) { const float4 t = red[j * G + tid]; m2.x += t.x; m2.y += t.y; m2.z += t.z; m2.w += t.w; }
        float* r0 = stats + ((size_t)blockIdx.x * 2) * cop + tid * 4;
        float* r1 = stats + ((size_t)blockIdx.x * 2 + 1) * cop + tid * 4;
        amx_st4(r0, tot); amx_st4(r1, m2);
    }
    // padded statistics columns [Cs, cop) stay untouched: the finalizer only reads the first C channels
}

extern "C" int amx_dropout_fwd(float* a, float* mask, const float* mask_in, float p, long seed, float* stats, long npix,
                               int Cs, int cop, int rows, int rows_pix, void* stream) {
    if (!a || !mask) AMX_BADARG(1);
    if (!(p > 0.f) || !(p < 1.f)) AMX_BADARG(2);
    if (npix <= 0 || Cs <= 0 || (Cs & 3) || Cs > 1024 || cop < Cs) AMX_BADARG(3);
    if (rows <= 0 || rows_pix <= 0 || (long)rows * rows_pix < npix) AMX_BADARG(4);
    AMX_LAUNCH(dropout_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, mask, mask_in, p, 1.0f / (1.0f - p),
               (unsigned)(seed & 0xffffffffL), (unsigned)((unsigned long long)seed >> 32), stats, npix, Cs, cop, rows_pix);
    AMX_CHECK_LAUNCH();
    return 0;
}

__global__ void dropout_bwd_kernel(float* __restrict__ d, const float* __restrict__ mask, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = amx_ld4(d + i * 4);
        const float4 m = amx_ld4(mask + i * 4);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        amx_st4(d + i * 4, v);
    }
}

// dpre *= mask in place (n elements, multiple of 4)
extern "C" int amx_dropout_bwd(float* dpre, const float* mask, long n, void* stream) {
    if (!dpre || !mask || n <= 0 || (n & 3)) AMX_BADARG(1);
    long nb = (n / 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    AMX_LAUNCH(dropout_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dpre, mask, n / 4);
    AMX_CHECK_LAUNCH();
    return 0;
}
