// linear.hip — dense layers on fp32 MFMA: C[M][N] = act(A[M][K] * B[K][N] + bias[N]).
//
// Replaces the nn.Linear (+ Tanh / ReLU) calls of the reference's dense nets, forward and backward:
//   fcEncoderNet / fcDecoderNet                     atomai/nets/ed.py:292-343, 530-580
//   the Linear heads of convEncoderNet              atomai/nets/ed.py:231-289
//   fcFeatureExtractor (DKL)                        atomai/nets/gp.py:14-26
// (ATen dispatches them to rocBLAS; they are < 1 % of the rVAE step's FLOPs, SURVEY.md §8-B1, but the plain VAE
// and the DKL extractor consist of nothing else.)
//
// One kernel serves the three GEMMs of a layer through operand strides:
//   forward   y  = x  * W^T  (+ b, act)    A = x  [M][K] (k contiguous),  B = W^T: element (k, n) = W[n][k] (k contiguous)
//   dgrad     dx = dpre * W                A = dpre [M][N'] (k contiguous), B = W [N'][K] (n contiguous)
//   wgrad     dW = dpre^T * x              A = dpre^T: (m, k) = dpre[k][m] (m contiguous), B = x [M'][K] (n contiguous)
// Mapping to CDNA4: 64 x 64 (or, for small problems, 32 x 32) output tile per workgroup, 4 waves in a 2 x 2 grid, each wave 2 x 2 (1 x 1) tiles of
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, k ascending -> deterministic, no split-K atomics); K is streamed in
// steps of 16 through LDS images [k/4][row][4] so that every operand fragment is one conflict-free ds_read_b128 (the
// layout of conv_kernel.h); the next step's operands are prefetched into registers under the current step's MFMAs.
// Operands that are contiguous along k and 16-byte aligned are fetched as float4; any other stride / alignment / edge
// goes through a scalar loader with per-element bounds (zero fill).
#include "amx_device.h"
#include <cstdlib>

#define LBK 16

struct GemmArgs {
    const float* A; long sam, sak;      // element (m, k) at A[m * sam + k * sak]
    const float* B; long sbk, sbn;      // element (k, n) at B[k * sbk + n * sbn]
    float* C; long scm;                 // element (m, n) at C[m * scm + n]
    const float* bias;                  // [N] or nullptr
    int M, N, K;
    int act;                            // 0 none, 1 tanh, 2 relu
    int vecA, vecB;                     // 1: k-contiguous, aligned float4 path is legal for this operand
    int kchunk;                         // split-K: k range of blockIdx.z is [z * kchunk, min(K, (z + 1) * kchunk)) (multiple
                                        // of 16); 0 = the whole K in one workgroup
    float* work;                        // split-K: raw partial sums [splits][M][N] (bias / activation in the reduce kernel)
};

// One operand tile (ROWS rows x 16 k) -> 4 registers per thread.  `rs` / `ks` are the row / k strides.
// Thread t (< 4 * ROWS) owns row = t >> 2, k-group kg = t & 3 (4 consecutive k): the LDS slot [kg][row][0..3].
static __device__ __forceinline__ float4 gemm_load(const float* P, long rs, long ks, int row0, int nrows, int k0, int K,
                                                   int vec, int tid) {
    const int row = row0 + (tid >> 2), k = k0 + (tid & 3) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float* p = P + (long)row * rs + (long)k * ks;
        if (vec && k + 3 < K) {
            v = amx_ld4(p);
        } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[ks];
            if (k + 2 < K) v.z = p[2 * ks];
            if (k + 3 < K) v.w = p[3 * ks];
        }
    }
    return v;
}

// TW x TW tiles of 16 x 16 per wave, 2 x 2 waves: output tile 32 TW x 32 TW per workgroup (TW = 2: 64 x 64; TW = 1:
// 32 x 32 for problems whose 64 x 64 grid would leave most of the 256 CUs idle, e.g. the 512 x 128 x 4096 first
// encoder layer of the VAEs: 16 -> 64 workgroups).
template <int TW>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
    constexpr int LB = 32 * TW;                                  // rows of A / columns of B per workgroup
    __shared__ __attribute__((aligned(16))) float sA[4 * LB * 4];
    __shared__ __attribute__((aligned(16))) float sB[4 * LB * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 waves
    const int m0 = blockIdx.y * LB, n0 = blockIdx.x * LB;
    const bool loader = tid < 4 * LB;                            // TW = 1: 128 threads fetch a 32-row operand tile

    f32x4 acc[TW][TW];
    #pragma unroll
    for (int i = 0; i < TW; ++i)
        #pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K (deterministic: every slice is a k-ascending chain, the slices are added in slice order by
    // gemm_splitk_reduce_kernel): a 512 x 128 x 4096 layer is 64 workgroups of 256 serial k steps otherwise (167 us)
    const int kbeg = a.kchunk ? (int)blockIdx.z * a.kchunk : 0;
    const int kend = a.kchunk ? (kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K) : a.K;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    if (loader) {
        ra = gemm_load(a.A, a.sam, a.sak, m0, a.M, kbeg, kend, a.vecA, tid);
        rb = gemm_load(a.B, a.sbn, a.sbk, n0, a.N, kbeg, kend, a.vecB, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += LBK) {
        if (loader) {
            amx_st4(sA + ((tid & 3) * LB + (tid >> 2)) * 4, ra);
            amx_st4(sB + ((tid & 3) * LB + (tid >> 2)) * 4, rb);
        }
        __syncthreads();
        if (loader && k0 + LBK < kend) {
            ra = gemm_load(a.A, a.sam, a.sak, m0, a.M, k0 + LBK, kend, a.vecA, tid);
            rb = gemm_load(a.B, a.sbn, a.sbk, n0, a.N, k0 + LBK, kend, a.vecB, tid);
        }
        float4 af[TW], bf[TW];
        #pragma unroll
        for (int i = 0; i < TW; ++i) af[i] = amx_ld4(sA + (g * LB + (wm * TW + i) * 16 + p) * 4);
        #pragma unroll
        for (int j = 0; j < TW; ++j) bf[j] = amx_ld4(sB + (g * LB + (wn * TW + j) * 16 + p) * 4);
        // one MFMA contracts k in {t, 4+t, 8+t, 12+t}; consecutive MFMAs target different accumulators
        #define AMX_GEMM_STEP(C_)                                                                                   \
            _Pragma("unroll") for (int i = 0; i < TW; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < TW; ++j)                                                      \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].C_, bf[j].C_, acc[i][j], 0, 0, 0);
        AMX_GEMM_STEP(x) AMX_GEMM_STEP(y) AMX_GEMM_STEP(z) AMX_GEMM_STEP(w)
        #undef AMX_GEMM_STEP
        __syncthreads();
    }
    // D fragment: column n = p, rows 4 g + r
    if (a.kchunk) {                                              // raw partial sums of this k slice
        float* W = a.work + (size_t)blockIdx.z * a.M * a.N;
        #pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int n = n0 + (wn * TW + j) * 16 + p;
            if (n >= a.N) continue;
            #pragma unroll
            for (int i = 0; i < TW; ++i)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * TW + i) * 16 + 4 * g + r;
                    if (m < a.M) W[(size_t)m * a.N + n] = acc[i][j][r];
                }
        }
        return;
    }
    #pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int n = n0 + (wn * TW + j) * 16 + p;
        if (n >= a.N) continue;
        const float b = a.bias ? a.bias[n] : 0.f;
        #pragma unroll
        for (int i = 0; i < TW; ++i)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TW + i) * 16 + 4 * g + r;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + b;
                if (a.act == 1) v = tanhf(v);
                else if (a.act == 2) v = v > 0.f ? v : 0.f;
                a.C[(long)m * a.scm + n] = v;
            }
    }
}

static int aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// C[m][n] = act(bias[n] + sum_z work[z][m][n]), z ascending
__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ work, int splits, float* __restrict__ C, long scm,
                                          const float* __restrict__ bias, int M, int N, int act) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N); const long m = i / N;
        float v = work[i];
        for (int z = 1; z < splits; ++z) v += work[(size_t)z * total + i];
        v += bias ? bias[n] : 0.f;
        if (act == 1) v = tanhf(v);
        else if (act == 2) v = v > 0.f ? v : 0.f;
        C[m * scm + n] = v;
    }
}

// C[M][N] = act(A * B + bias) with arbitrary operand strides (see the file header for the three uses).
// Number of k slices amx_gemm_f32_splitk will use for this problem (1 = no split: plain amx_gemm_f32); the caller
// provides splits * M * N floats of workspace.
extern "C" int amx_gemm_f32_splits(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const long wg = (long)amx_ceil_div(N, 32) * amx_ceil_div(M, 32);     // 32 x 32 tiles (the small-problem plan)
    if (wg >= 128 || K < 1024) return 1;
    int s = (int)((512 + wg - 1) / wg);                          // aim at ~512 workgroups
    if (s > K / 256) s = K / 256;                                // >= 256 k per slice
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

extern "C" int amx_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                                   long scm, const float* bias, int M, int N, int K, int act, float* work, int splits,
                                   void* stream);

extern "C" int amx_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                            long scm, const float* bias, int M, int N, int K, int act, void* stream) {
    return amx_gemm_f32_splitk(A, sam, sak, B, sbk, sbn, C, scm, bias, M, N, K, act, nullptr, 1, stream);
}

extern "C" int amx_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                                   long scm, const float* bias, int M, int N, int K, int act, float* work, int splits,
                                   void* stream) {
    if (!A || !B || !C) AMX_BADARG(1);
    if (M <= 0 || N <= 0 || K <= 0) AMX_BADARG(2);
    if (act < 0 || act > 2 || scm < N) AMX_BADARG(3);
    if (splits < 1 || splits > 64 || (splits > 1 && !work)) AMX_BADARG(5);
    GemmArgs a;
    a.kchunk = 0; a.work = nullptr;
    a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.scm = scm; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.act = act;
    a.vecA = sak == 1 && (sam & 3) == 0 && aligned16(A);
    a.vecB = sbk == 1 && (sbn & 3) == 0 && aligned16(B);
    // 64 x 64 tiles unless they would leave most of the chip idle (< 128 workgroups): then 32 x 32 (4x the workgroups)
    const long wg64 = (long)amx_ceil_div(N, 64) * amx_ceil_div(M, 64);
    int tile = wg64 < 128 ? 32 : 64;
    { const int v = amx_knobs().gemm_tile; if (v == 32 || v == 64) tile = v; }                          // AMX_GEMM_TILE (A/B)
    dim3 grid(amx_ceil_div(N, tile), amx_ceil_div(M, tile));
    if (grid.y > 65535) AMX_BADARG(4);
    if (splits > 1) {
        a.kchunk = amx_round_up(amx_ceil_div(K, splits), LBK);
        a.work = work;
        grid.z = amx_ceil_div(K, a.kchunk);
        tile = 32; grid.x = amx_ceil_div(N, 32); grid.y = amx_ceil_div(M, 32);
    }
    if (tile == 32) AMX_LAUNCH(gemm_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else AMX_LAUNCH(gemm_f32_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    AMX_CHECK_LAUNCH();
    if (splits > 1) {
        long nb = ((long)M * N + 255) / 256;
        if (nb > 4096) nb = 4096;
        AMX_LAUNCH(gemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, work, (int)grid.z, C,
                   scm, bias, M, N, act);
        AMX_CHECK_LAUNCH();
    }
    return 0;
}

// dpre = dy * act'(y) for the activation fused into the forward epilogue (y is the layer OUTPUT):
// tanh: 1 - y^2, relu: y > 0.  n elements, any alignment.
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                               long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float yy = y[i], d = dy[i];
        out[i] = act == 1 ? d * (1.f - yy * yy) : (yy > 0.f ? d : 0.f);
    }
}

extern "C" int amx_act_bwd(const float* dy, const float* y, float* out, long n, int act, void* stream) {
    if (!dy || !y || !out || n <= 0) AMX_BADARG(1);
    if (act != 1 && act != 2) AMX_BADARG(2);
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    AMX_LAUNCH(act_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, y, out, n, act);
    AMX_CHECK_LAUNCH();
    return 0;
}
