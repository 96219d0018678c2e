// kernel_matrix.hip — dense, tiled covariance evaluation for deep-kernel-learning GP regression.
//
//   RBF-ARD      k(x,x') = s2 * exp(-1/2 sum_d ((x_d - x'_d)/l_d)^2)
//   Matern-5/2   k(x,x') = s2 * (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r),  r = ||(x - x')/l||
// evaluated on the feature extractor's embeddings (reference: ScaleKernel(RBFKernel(ard)) /
// MaternKernel inside gpytorch, selected at atomai/nets/gp.py:41-46, 95-106; the arithmetic lives in the
// un-vendored dependency gpytorch>=1.9.1 — closed forms restated from its documentation).
//
// Three entry points, all for fp32 and fp64 (dklGPR defaults to double precision, gptrainer.py:172-173):
//   amx_kernel_matrix      K[N][M] (+ noise on the diagonal)          HBM-write bound: 16 B/lane coalesced rows
//   amx_kernel_matvec      Y = K(X1,X2) V without materialising K     (predictive means / CG-type solvers)
//   amx_kernel_matrix_bwd  given G = dL/dK (symmetric, X1 == X2): dX, d(1/l), d(s2) by recomputing K tile-wise;
//                          per-workgroup partial rows, wave-level shuffles, deterministic.
#include "amx_device.h"
#include <cstdlib>

#define KM_MAXD 16
// rows of K per workgroup of the builder: measured (tools/gpu_km_ab.py, 16384^2): fp32 32 rows 5.6 TB/s / 64 rows 5.0;
// fp64 32 rows 4.1 / 64 rows 4.7
template <typename T> struct KmRows { static constexpr int value = sizeof(T) == 4 ? 32 : 64; };

template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<double> { static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ T km_exp(T x);
template <> __device__ __forceinline__ float km_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double km_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T km_fma(T a, T b, T c);
template <> __device__ __forceinline__ float km_fma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double km_fma<double>(double a, double b, double c) { return fma(a, b, c); }
template <typename T> __device__ __forceinline__ T km_sqrt(T x);
template <> __device__ __forceinline__ float km_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double km_sqrt<double>(double x) { return sqrt(x); }

// value and radial factor:  k = s2 * f(r2);  returns also w with  dk/d(diff_d) = -w * s_d^2 * diff_d
// fp32 builder only: exp on the hardware transcendental unit (v_exp_f32 = 2^x, ~1 ulp; the argument product adds
// |x| * 9e-8 relative error — below the fp32 tolerance of the path for every |x| that does not underflow anyway).
// The library expf costs ~20 VALU instructions per element, which at 2.7e8 elements is as long as the 1 GB write.
static __device__ __forceinline__ float km_fast_exp(float x) {
#ifdef AMX_EMU
    return expf(x);
#else
    return __expf(x);
#endif
}

// exp(x) for the fp64 builder, x <= 0 (the RBF / Matern arguments): k = round(x log2 e), r = x - k ln 2 in two pieces
// (|r| <= 0.347), a degree-13 Taylor polynomial in Horner form (remainder r^14 / 14! < 5e-18: below half an ulp), scaled by
// 2^k with v_ldexp_f64 (denormal results and the underflow to 0 below x = -745 come out of the instruction).  ~20 fp64
// instructions and no branches; the device library's exp was ~2/3 of the instructions of an fp64 covariance element
// (round 6: 0.613 -> see profiles/r06_bench_extra.log).  Relative error <= 2 ulp (tests/_gp_checks.py compares the
// builder with numpy's exp in fp64 at 1e-14).
static __device__ __forceinline__ double km_exp_neg(double x) {
#ifdef AMX_EMU
    return exp(x);
#else
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);          // ln2 high part (32 trailing zero bits)
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);                   // ln2 low part
    // Horner steps as v_fma_f64 with the coefficient in an SGPR pair: the compiler's choice, v_fmac_f64 (d += a * b), needs a
    // v_mov_b64 of the coefficient into the destination before every step — 13 moves per element
    double p = 1.0 / 6227020800.0;                                          // 1/13!
#define KM_HORNER(c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"((double)(c)))
    KM_HORNER(1.0 / 479001600.0); KM_HORNER(1.0 / 39916800.0); KM_HORNER(1.0 / 3628800.0); KM_HORNER(1.0 / 362880.0);
    KM_HORNER(1.0 / 40320.0); KM_HORNER(1.0 / 5040.0); KM_HORNER(1.0 / 720.0); KM_HORNER(1.0 / 120.0);
    KM_HORNER(1.0 / 24.0); KM_HORNER(1.0 / 6.0); KM_HORNER(0.5); KM_HORNER(1.0); KM_HORNER(1.0);
#undef KM_HORNER
    const double kk = k < -2000.0 ? -2000.0 : k;                            // (the int conversion must not overflow)
    return __builtin_amdgcn_ldexp(p, (int)kk);
#endif
}

template <typename T>
__device__ __forceinline__ T km_eval(T r2, T s2, int kind, T* w) {
    if (kind == 0) {
        const T k = s2 * km_exp<T>(T(-0.5) * r2);
        *w = k;
        return k;
    }
    const T sq5 = T(2.23606797749978969641);
    const T r = km_sqrt<T>(r2);
    const T e = km_exp<T>(-sq5 * r);
    *w = s2 * T(5.0 / 3.0) * (T(1) + sq5 * r) * e;
    return s2 * (T(1) + sq5 * r + T(5.0 / 3.0) * r2) * e;
}

// ------------------------------------------------------------------ K = k(X1, X2) [+ noise * I]
// DR: register-resident embedding dimensions of a thread's columns (4, 8 or 16 >= D).
// Round 6 (D = 8 ran at 0.40 of the HBM peak, VALU-bound): the dimension loop is BRANCH-FREE — both
// coordinate images are zero-padded to DR, so the DR - D extra terms add exact zeros instead of costing a scalar
// compare-and-branch per dimension and column (the D = 8 instantiation had 246 basic blocks) —, the squared distance is an
// fma chain (d ascending, as before: at most one rounding per term less), the kernel family and the exp flavour are
// template parameters, and in fp32 a thread's four columns are two PAIRS held in <2 x float> registers so that the
// subtraction and the fma of two columns are ONE packed instruction each (v_pk_add_f32 / v_pk_fma_f32: 2 instead of 4 VALU
// operations per dimension and column pair); a row's coordinates come from LDS as broadcast 16-byte reads.
#ifndef AMX_EMU
typedef float km_f2 __attribute__((ext_vector_type(2)));
#define KM_PACKED 1
#else
#define KM_PACKED 0        // (the CPU emulator build is plain g++: same fma chain per column, unpacked)
#endif

template <typename T, int DR, int KIND, bool FAST>
__global__ __launch_bounds__(256) void kernel_matrix_kernel(const T* __restrict__ X1, const T* __restrict__ X2,
                                                            const T* __restrict__ inv_ls, T s2,
                                                            T noise, int N, int M, int D, T* __restrict__ K,
                                                            int stream_stores) {
    constexpr int V = Vec16<T>::N;
    constexpr int COLS = 64 * V;                         // columns per workgroup
    constexpr int KM_ROWS = KmRows<T>::value;
    __shared__ __attribute__((aligned(16))) T s_x1[KM_ROWS * DR];    // [row][DR], zero beyond D
    __shared__ T s_x2[DR * COLS];                        // [d][col]
    const int tid = threadIdx.x;
    const int row0 = blockIdx.y * KM_ROWS, col0 = blockIdx.x * COLS;
    for (int i = tid; i < KM_ROWS * DR; i += 256) {
        const int r = i / DR, d = i - r * DR;
        s_x1[i] = (d < D && row0 + r < N) ? X1[(size_t)(row0 + r) * D + d] * inv_ls[d] : T(0);
    }
    for (int i = tid; i < COLS * D; i += 256) {
        const int c = i / D, d = i - c * D;
        s_x2[d * COLS + c] = col0 + c < M ? X2[(size_t)(col0 + c) * D + d] * inv_ls[d] : T(0);
    }
    __syncthreads();
    const int cg = tid & 63, rg = tid >> 6;              // 64 column groups x 4 row groups
    // a thread's V columns are the same for every row it writes: their scaled coordinates live in registers
    T x2r[V][DR];
    #pragma unroll
    for (int v = 0; v < V; ++v)
        #pragma unroll
        for (int d = 0; d < DR; ++d) x2r[v][d] = d < D ? s_x2[d * COLS + cg * V + v] : T(0);
    const int gc = col0 + cg * V;
    const bool vec_ok = gc + V <= M && ((M * sizeof(T)) % 16 == 0);
    #pragma unroll 2
    for (int rr = 0; rr < KM_ROWS / 4; ++rr) {
        const int r = rg * (KM_ROWS / 4) + rr;
        const int gi = row0 + r;
        if (gi >= N) break;
        T x1[DR];
        #pragma unroll
        for (int d = 0; d < DR; ++d) x1[d] = s_x1[r * DR + d];              // (broadcast reads, merged to b128 / b64)
        T r2[V];
#if KM_PACKED
        if constexpr (sizeof(T) == 4) {
            #pragma unroll
            for (int j = 0; j < V / 2; ++j) {
                km_f2 acc = {0.f, 0.f};
                #pragma unroll
                for (int d = 0; d < DR; ++d) {
                    const km_f2 xc = {(float)x2r[2 * j][d], (float)x2r[2 * j + 1][d]};
                    const km_f2 xr = {(float)x1[d], (float)x1[d]};
                    const km_f2 df = xr - xc;
                    acc = __builtin_elementwise_fma(df, df, acc);
                }
                r2[2 * j] = acc.x; r2[2 * j + 1] = acc.y;
            }
        } else
#endif
        {
            #pragma unroll
            for (int v = 0; v < V; ++v) {
                T acc = T(0);
                #pragma unroll
                for (int d = 0; d < DR; ++d) { const T df = x1[d] - x2r[v][d]; acc = km_fma<T>(df, df, acc); }
                r2[v] = acc;
            }
        }
        T out[V];
        #pragma unroll
        for (int v = 0; v < V; ++v) {
            T w;
            T k;
            if constexpr (sizeof(T) == 4 && KIND == 0 && FAST) k = (T)((float)s2 * km_fast_exp(-0.5f * (float)r2[v]));
            else if constexpr (sizeof(T) == 8 && KIND == 0 && FAST) k = (T)((double)s2 * km_exp_neg(-0.5 * (double)r2[v]));
            else if constexpr (sizeof(T) == 8 && KIND == 1 && FAST) {
                const double sq5 = 2.23606797749978969641, r = sqrt((double)r2[v]);
                k = (T)((double)s2 * (1.0 + sq5 * r + (5.0 / 3.0) * (double)r2[v]) * km_exp_neg(-sq5 * r));
            }
            else k = km_eval<T>(r2[v], s2, KIND, &w);
            if (gi == gc + v) k += noise;
            out[v] = k;
        }
        T* dst = K + (size_t)gi * M + gc;
        if (vec_ok) {
            if constexpr (V == 4) {
                if (stream_stores) amx_st4_stream(reinterpret_cast<float*>(dst), *reinterpret_cast<float4*>(out));
                else *reinterpret_cast<float4*>(dst) = *reinterpret_cast<float4*>(out);
            } else { dst[0] = out[0]; dst[1] = out[1]; }
        } else {
            #pragma unroll
            for (int v = 0; v < V; ++v) if (gc + v < M) dst[v] = out[v];
        }
    }
}

template <typename T, int DR>
static int launch_km_dr(const void* X1, const void* X2, const void* inv_ls, double s2, int kind, double noise,
                        int N, int M, int D, void* K, hipStream_t st) {
    constexpr int COLS = 64 * Vec16<T>::N;
    dim3 grid(amx_ceil_div(M, COLS), amx_ceil_div(N, KmRows<T>::value));
    // streaming stores (+6..9 %) and the hardware exp of the fp32 RBF builder: profiles/r02_logs/r02w_km_ab.log
    if (kind == 0) {
        AMX_LAUNCH((kernel_matrix_kernel<T, DR, 0, true>), grid, dim3(256), 0, st, (const T*)X1, (const T*)X2,
                   (const T*)inv_ls, (T)s2, (T)noise, N, M, D, (T*)K, 1);
    } else {
        AMX_LAUNCH((kernel_matrix_kernel<T, DR, 1, sizeof(T) == 8>), grid, dim3(256), 0, st, (const T*)X1, (const T*)X2,
                   (const T*)inv_ls, (T)s2, (T)noise, N, M, D, (T*)K, 1);
    }
    AMX_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int launch_km(const void* X1, const void* X2, const void* inv_ls, double s2, int kind, double noise,
                     int N, int M, int D, void* K, hipStream_t st) {
    if (D <= 4) return launch_km_dr<T, 4>(X1, X2, inv_ls, s2, kind, noise, N, M, D, K, st);
    if (D <= 8) return launch_km_dr<T, 8>(X1, X2, inv_ls, s2, kind, noise, N, M, D, K, st);
    return launch_km_dr<T, 16>(X1, X2, inv_ls, s2, kind, noise, N, M, D, K, st);
}

extern "C" int amx_kernel_matrix(const void* X1, const void* X2, const void* inv_ls, double outputscale,
                                 int kind, double noise, int N, int M, int D, int is_double, void* K,
                                 void* stream) {
    if (!X1 || !X2 || !inv_ls || !K) AMX_BADARG(1);
    if (N <= 0 || M <= 0 || D <= 0 || D > KM_MAXD || (kind != 0 && kind != 1)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    return is_double ? launch_km<double>(X1, X2, inv_ls, outputscale, kind, noise, N, M, D, K, st)
                     : launch_km<float>(X1, X2, inv_ls, outputscale, kind, noise, N, M, D, K, st);
}

// ------------------------------------------------------------------ Y[N][R] = K(X1, X2) V[M][R]
#define KM_MAXR 4
template <typename T>
__global__ __launch_bounds__(256) void kernel_matvec_kernel(const T* __restrict__ X1, const T* __restrict__ X2,
                                                            const T* __restrict__ inv_ls, T s2, int kind, int N,
                                                            int M, int D, int R, const T* __restrict__ Vv,
                                                            T* __restrict__ Y) {
    // one wave per row: lanes stride over the columns, then a wave-level reduction
    __shared__ T s_x1[4 * KM_MAXD];
    __shared__ T s_t[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = blockIdx.x * 4 + wave;
    if (lane < D && gi < N) s_x1[wave * KM_MAXD + lane] = X1[(size_t)gi * D + lane] * inv_ls[lane];
    __syncthreads();
    T acc[KM_MAXR];
    #pragma unroll
    for (int q = 0; q < KM_MAXR; ++q) acc[q] = T(0);
    if (gi < N)
        for (int c = lane; c < M; c += 64) {
            T r2 = T(0);
            for (int d = 0; d < D; ++d) {
                const T df = s_x1[wave * KM_MAXD + d] - X2[(size_t)c * D + d] * inv_ls[d];
                r2 += df * df;
            }
            T w;
            const T k = km_eval<T>(r2, s2, kind, &w);
            #pragma unroll
            for (int q = 0; q < KM_MAXR; ++q) if (q < R) acc[q] += k * Vv[(size_t)c * R + q];
        }
    #pragma unroll
    for (int q = 0; q < KM_MAXR; ++q) {
        if (q >= R) break;
        // per-wave tree through LDS: identical code for fp32 and fp64, fixed order -> deterministic
        s_t[tid] = acc[q];
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { if (lane < o) s_t[tid] += s_t[tid + o]; __syncthreads(); }
        if (lane == 0 && gi < N) Y[(size_t)gi * R + q] = s_t[tid];
        __syncthreads();
    }
}

template <typename T>
static int launch_mv(const void* X1, const void* X2, const void* inv_ls, double s2, int kind, int N, int M, int D,
                     int R, const void* V, void* Y, hipStream_t st) {
    AMX_LAUNCH(kernel_matvec_kernel<T>, dim3(amx_ceil_div(N, 4)), dim3(256), 0, st, (const T*)X1, (const T*)X2,
               (const T*)inv_ls, (T)s2, kind, N, M, D, R, (const T*)V, (T*)Y);
    AMX_CHECK_LAUNCH();
    return 0;
}

extern "C" int amx_kernel_matvec(const void* X1, const void* X2, const void* inv_ls, double outputscale, int kind,
                                 int N, int M, int D, int R, int is_double, const void* V, void* Y, void* stream) {
    if (!X1 || !X2 || !inv_ls || !V || !Y) AMX_BADARG(1);
    if (N <= 0 || M <= 0 || D <= 0 || D > KM_MAXD || R < 1 || R > KM_MAXR || (kind != 0 && kind != 1)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    return is_double ? launch_mv<double>(X1, X2, inv_ls, outputscale, kind, N, M, D, R, V, Y, st)
                     : launch_mv<float>(X1, X2, inv_ls, outputscale, kind, N, M, D, R, V, Y, st);
}

// ------------------------------------------------------------------ backward (X1 == X2, G symmetric)
// dX[i][d]    = 2 * sum_j G_ij * dk_ij/dx_id        = -2 s_d^2 sum_j G_ij w_ij (x_id - x_jd)
// dinv_ls[d]  = sum_ij G_ij * dk_ij/ds_d            = -s_d sum_ij G_ij w_ij (x_id - x_jd)^2      (s = 1/l)
// ds2         = sum_ij G_ij k_ij / s2                                                             (noise-free k)
// One wave per row i; partial rows [N/4 blocks][D + 1] for (dinv_ls, ds2), reduced by amx_reduce_rows-style host call.
template <typename T>
__global__ __launch_bounds__(256) void kernel_matrix_bwd_kernel(const T* __restrict__ X, const T* __restrict__ inv_ls,
                                                                T s2, int kind, int N, int D,
                                                                const T* __restrict__ G, T* __restrict__ dX,
                                                                T* __restrict__ part,
                                                                const T* __restrict__ alpha, T gscale) {
    // alpha != nullptr (amx_kernel_matrix_bwd_mll): G holds K^-1 and the upstream gradient of the exact marginal log
    // likelihood, (alpha_i alpha_j - K^-1_ij) * gscale, is formed while reading it — no N x N temporary for
    // alpha alpha^T, the difference or the scaling (nets/gp.py:_ExactMLLFn.backward)
    __shared__ T s_t[256];
    __shared__ T s_acc[4][KM_MAXD + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = blockIdx.x * 4 + wave;
    T xi[KM_MAXD], sl[KM_MAXD];
    #pragma unroll
    for (int d = 0; d < KM_MAXD; ++d) { xi[d] = T(0); sl[d] = T(0); }
    for (int d = 0; d < D; ++d) { sl[d] = inv_ls[d]; if (gi < N) xi[d] = X[(size_t)gi * D + d]; }
    T ax[KM_MAXD], al[KM_MAXD], as2 = T(0);
    #pragma unroll
    for (int d = 0; d < KM_MAXD; ++d) { ax[d] = T(0); al[d] = T(0); }
    const T ai = (alpha && gi < N) ? alpha[gi] : T(0);
    if (gi < N)
        for (int c = lane; c < N; c += 64) {
            T df[KM_MAXD];
            T r2 = T(0);
            #pragma unroll
            for (int d = 0; d < KM_MAXD; ++d) {
                if (d >= D) break;
                df[d] = xi[d] - X[(size_t)c * D + d];
                const T u = df[d] * sl[d];
                r2 += u * u;
            }
            T w;
            const T k = km_eval<T>(r2, s2, kind, &w);
            T g = G[(size_t)gi * N + c];
            if (alpha) g = (ai * alpha[c] - g) * gscale;
            as2 += g * k;
            const T gw = g * w;
            #pragma unroll
            for (int d = 0; d < KM_MAXD; ++d) {
                if (d >= D) break;
                ax[d] += gw * df[d];
                al[d] += gw * df[d] * df[d];
            }
        }
    // wave reductions through LDS (works for fp32 and fp64 alike), fixed order -> deterministic
    for (int q = 0; q < 2 * D + 1; ++q) {
        T v = q < D ? ax[0] : (q < 2 * D ? al[0] : as2);
        #pragma unroll
        for (int d = 1; d < KM_MAXD; ++d) {
            if (q < D && q == d) v = ax[d];
            if (q == D + d && d < D) v = al[d];
        }
        s_t[tid] = v;
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { if (lane < o) s_t[tid] += s_t[tid + o]; __syncthreads(); }
        if (lane == 0) {
            const T tot = s_t[tid];
            if (q < D) { if (gi < N) dX[(size_t)gi * D + q] = T(-2) * sl[q] * sl[q] * tot; }
            else if (q < 2 * D) s_acc[wave][q - D] = gi < N ? -sl[q - D] * tot : T(0);
            else s_acc[wave][D] = gi < N ? tot / s2 : T(0);
        }
        __syncthreads();
    }
    if (tid <= D) part[(size_t)blockIdx.x * (D + 1) + tid] = s_acc[0][tid] + s_acc[1][tid] + s_acc[2][tid] + s_acc[3][tid];
}

template <typename T>
static int launch_kb(const void* X, const void* inv_ls, double s2, int kind, int N, int D, const void* G, void* dX,
                     void* part, hipStream_t st, const void* alpha = nullptr, double gscale = 1.0) {
    AMX_LAUNCH(kernel_matrix_bwd_kernel<T>, dim3(amx_ceil_div(N, 4)), dim3(256), 0, st, (const T*)X,
               (const T*)inv_ls, (T)s2, kind, N, D, (const T*)G, (T*)dX, (T*)part, (const T*)alpha, (T)gscale);
    AMX_CHECK_LAUNCH();
    return 0;
}

// part: [ceil(N/4)][D+1] partial rows of (d inv_ls[0..D), d outputscale)
extern "C" int amx_kernel_matrix_bwd(const void* X, const void* inv_ls, double outputscale, int kind, int N, int D,
                                     int is_double, const void* G, void* dX, void* part, void* stream) {
    if (!X || !inv_ls || !G || !dX || !part) AMX_BADARG(1);
    if (N <= 0 || D <= 0 || D > KM_MAXD || (kind != 0 && kind != 1)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    return is_double ? launch_kb<double>(X, inv_ls, outputscale, kind, N, D, G, dX, part, st)
                     : launch_kb<float>(X, inv_ls, outputscale, kind, N, D, G, dX, part, st);
}

// The same contraction with G = (alpha alpha^T - Kinv) * gscale formed on the fly from Kinv (symmetric, as
// torch.cholesky_inverse returns it) and alpha = K^-1 (y - mean): the backward pass of the exact marginal log likelihood
// (gscale = 0.5 / N) reads ONE N x N matrix instead of building three.
extern "C" int amx_kernel_matrix_bwd_mll(const void* X, const void* inv_ls, double outputscale, int kind, int N, int D,
                                         int is_double, const void* Kinv, const void* alpha, double gscale, void* dX,
                                         void* part, void* stream) {
    if (!X || !inv_ls || !Kinv || !alpha || !dX || !part) AMX_BADARG(1);
    if (N <= 0 || D <= 0 || D > KM_MAXD || (kind != 0 && kind != 1)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    return is_double ? launch_kb<double>(X, inv_ls, outputscale, kind, N, D, Kinv, dX, part, st, alpha, gscale)
                     : launch_kb<float>(X, inv_ls, outputscale, kind, N, D, Kinv, dX, part, st, alpha, gscale);
}
