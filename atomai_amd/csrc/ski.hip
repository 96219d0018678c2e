// ski.hip — structured kernel interpolation (KISS-GP) around the covariance builder: the reference's GP layer is
//   gpytorch.kernels.GridInterpolationKernel(ScaleKernel(RBFKernel(ard)), num_dims=embedim, grid_size=50)
// (atomai/nets/gp.py:41-46; trained through ExactMarginalLogLikelihood, atomai/trainers/gptrainer.py:126-137,303),
//   K_SKI(X, X') = W(X) K_UU W(X')^T,      K_UU = base kernel on a regular grid U of G points per embedding dimension,
//   W = local cubic-convolution interpolation weights (Keys 1981, a = -1/2): 4 grid nodes per dimension, 4^D per point.
// gpytorch (>= 1.9.1, setup.py:40) is not vendored: the conventions restated here are those of its published source
// (gpytorch/utils/interpolation.py Interpolation.interpolate, gpytorch/utils/grid.py create_grid) — PARITY UNPINNED, as
// the whole DKL row.
//
// Everything N-sized of a KISS-GP training step / prediction reduces to G^D-sized dense algebra (nets/gp.py) around
//   A = W^T W   (m x m, m = G^D),   b = W^T r   (r = y - mean),
// and their backward.  The kernels of this file (D = 1 or 2, fp32 and fp64; no float atomics — sums run in a fixed order
// over points sorted by grid cell, so two runs are bit-identical like the rest of the build):
//   amx_ski_weights     Z -> first stencil node, the 4 weights and their derivatives per point and dimension
//   amx_ski_gram        A and b: per-cell S x S blocks (S = 4^D) over the cell's points, then a gather of the <= 4^D cells
//                       that contain both nodes of an entry (the band of A; the caller hands A in zeroed)
//   amx_ski_gram_bwd    dL/dZ and dL/dr from dL/dA (symmetric) and dL/db: one wave per point, 4^D x 4^D gathers
//   amx_ski_interp      Y = W V for node vectors V (predictive means)
//   amx_ski_cov         scale * W1 Q W2^T (full or diagonal: predictive covariances / variances)
#include "amx_device.h"

#define SKI_MAXC 8          // right-hand sides (outputs sharing the embedding) per call

template <typename T> __device__ __forceinline__ T ski_floor(T x);
template <> __device__ __forceinline__ float ski_floor<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double ski_floor<double>(double x) { return floor(x); }

// Keys' cubic convolution kernel (a = -1/2) and its derivative at signed distance t (in grid spacings):
//   |t| <= 1: (1.5 |t| - 2.5) t^2 + 1;   1 < |t| <= 2: ((-0.5 |t| + 2.5) |t| - 4) |t| + 2;   0 beyond
template <typename T>
__device__ __forceinline__ void ski_keys(T t, T* w, T* dw) {
    const T u = t < T(0) ? -t : t, sg = t < T(0) ? T(-1) : T(1);
    if (u <= T(1)) { *w = (T(1.5) * u - T(2.5)) * u * u + T(1); *dw = sg * (T(4.5) * u - T(5)) * u; }
    else if (u <= T(2)) { *w = ((T(-0.5) * u + T(2.5)) * u - T(4)) * u + T(2); *dw = sg * ((T(-1.5) * u + T(5)) * u - T(4)); }
    else { *w = T(0); *dw = T(0); }
}

// ------------------------------------------------------------------ interpolation weights
// thread per (point, dimension): lower node L = floor((z - g0) / delta) (clamped so that the stencil L-1 .. L+2 stays on
// the grid), stencil node k at signed distance (z - g0) / delta - (L - 1 + k)
template <typename T>
__global__ __launch_bounds__(256) void ski_weights_kernel(const T* __restrict__ Z, const T* __restrict__ g0,
                                                          const T* __restrict__ inv_delta, int N, int D, int G,
                                                          int* __restrict__ base, T* __restrict__ w,
                                                          T* __restrict__ dw, int* __restrict__ cell) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int d = i % D;
    const T t = (Z[i] - g0[d]) * inv_delta[d];
    int L = (int)ski_floor<T>(t);
    L = L < 1 ? 1 : (L > G - 3 ? G - 3 : L);
    base[i] = L - 1;
    if (cell && d == 0) {                        // grid cell of the point = base[0] * (G - 3) + base[1] (the sort key of amx_ski_gram)
        int c = L - 1;
        if (D == 2) {
            int L1 = (int)ski_floor<T>((Z[i + 1] - g0[1]) * inv_delta[1]);
            L1 = L1 < 1 ? 1 : (L1 > G - 3 ? G - 3 : L1);
            c = c * (G - 3) + L1 - 1;
        }
        cell[i / D] = c;
    }
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        T wv, dv;
        ski_keys<T>(t - (T)(L - 1 + k), &wv, &dv);
        w[(size_t)i * 4 + k] = wv;
        dw[(size_t)i * 4 + k] = dv * inv_delta[d];
    }
}

// weight / node of stencil entry a of point n.  D = 2: a = 4 ax + ay (dimension 0 is the slow axis of the node index)
template <typename T, int DIM>
__device__ __forceinline__ T ski_omega(const T* __restrict__ w, int n, int a) {
    if (DIM == 1) return w[(size_t)n * 4 + a];
    return w[((size_t)n * 2) * 4 + (a >> 2)] * w[((size_t)n * 2 + 1) * 4 + (a & 3)];
}
template <int DIM>
__device__ __forceinline__ int ski_node(const int* __restrict__ base, int n, int a, int G) {
    if (DIM == 1) return base[n] + a;
    return (base[2 * n] + (a >> 2)) * G + base[2 * n + 1] + (a & 3);
}

// ------------------------------------------------------------------ A = W^T W, b = W^T r
// (1) one workgroup per grid cell (= stencil position, (G - 3)^D of them): thread (a, a') adds omega_a omega_a' over the
//     cell's points in sorted order; threads a' == 0 also add omega_a r_c.   blocks [cell][S][S], bvec [cell][C][S]
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_cell_blocks_kernel(const T* __restrict__ w, const T* __restrict__ r,
                                                              const int* __restrict__ order,
                                                              const int* __restrict__ cell_start, int N, int C,
                                                              T* __restrict__ blocks, T* __restrict__ bvec) {
    constexpr int S = DIM == 1 ? 4 : 16;
    const int cell = blockIdx.x, tid = threadIdx.x;
    if (tid >= S * S) return;
    const int a = tid / S, a2 = tid - a * S;
    const int p0 = cell_start[cell], p1 = cell_start[cell + 1];
    T acc = T(0);
    T accb[SKI_MAXC];
    #pragma unroll
    for (int c = 0; c < SKI_MAXC; ++c) accb[c] = T(0);
    for (int p = p0; p < p1; ++p) {
        const int n = order[p];
        const T wa = ski_omega<T, DIM>(w, n, a);
        acc += wa * ski_omega<T, DIM>(w, n, a2);
        if (a2 == 0)
            for (int c = 0; c < C; ++c) accb[c] += wa * r[(size_t)c * N + n];
    }
    blocks[(size_t)cell * S * S + tid] = acc;
    if (a2 == 0)
        for (int c = 0; c < C; ++c) bvec[((size_t)cell * C + c) * S + a] = accb[c];
}

// (2) A [m][m], ZERO on entry: thread per (u, offset) over the band — nodes further apart than 3 in any dimension share
//     no cell, those entries are never touched (7^D of m entries per row: the dense writer spent 98 % of its threads on zeros)
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_gram_gather_kernel(const T* __restrict__ blocks, int G, T* __restrict__ A) {
    constexpr int S = DIM == 1 ? 4 : 16;
    constexpr int NOFF = DIM == 1 ? 7 : 49;
    const int m = DIM == 1 ? G : G * G, NC = G - 3;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m * NOFF) return;
    const int u = e / NOFF, o = e - u * NOFF;
    T acc = T(0);
    int v;
    if (DIM == 1) {
        v = u + o - 3;
        if (v < 0 || v >= G) return;
        const int lo = max(max(u, v) - 3, 0), hi = min(min(u, v), NC - 1);
        for (int c = lo; c <= hi; ++c) acc += blocks[((size_t)c * S + (u - c)) * S + (v - c)];
    } else {
        const int ux = u / G, uy = u - ux * G, vx = ux + o / 7 - 3, vy = uy + o % 7 - 3;
        if (vx < 0 || vx >= G || vy < 0 || vy >= G) return;
        v = vx * G + vy;
        const int xlo = max(max(ux, vx) - 3, 0), xhi = min(min(ux, vx), NC - 1);
        const int ylo = max(max(uy, vy) - 3, 0), yhi = min(min(uy, vy), NC - 1);
        for (int cx = xlo; cx <= xhi; ++cx)
            for (int cy = ylo; cy <= yhi; ++cy) {
                const int a = (ux - cx) * 4 + (uy - cy), a2 = (vx - cx) * 4 + (vy - cy);
                acc += blocks[((size_t)(cx * NC + cy) * S + a) * S + a2];
            }
    }
    A[(size_t)u * m + v] = acc;
}

// (3) b [C][m]: thread per (c, u)
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_bvec_gather_kernel(const T* __restrict__ bvec, int G, int C, T* __restrict__ b) {
    constexpr int S = DIM == 1 ? 4 : 16;
    const int m = DIM == 1 ? G : G * G, NC = G - 3;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * m) return;
    const int c = e / m, u = e - c * m;
    T acc = T(0);
    if (DIM == 1) {
        for (int cc = max(u - 3, 0); cc <= min(u, NC - 1); ++cc) acc += bvec[((size_t)cc * C + c) * S + (u - cc)];
    } else {
        const int ux = u / G, uy = u - ux * G;
        for (int cx = max(ux - 3, 0); cx <= min(ux, NC - 1); ++cx)
            for (int cy = max(uy - 3, 0); cy <= min(uy, NC - 1); ++cy)
                acc += bvec[((size_t)(cx * NC + cy) * C + c) * S + (ux - cx) * 4 + (uy - cy)];
    }
    b[e] = acc;
}

// ------------------------------------------------------------------ backward of (A, b) w.r.t. the points
// L depends on Z through omega_n:  dL/domega_na = 2 sum_a' GA[u_a][u_a'] omega_na' + sum_c gb[c][u_a] r_cn   (GA symmetric)
//   dZ[n][d] = sum_a dL/domega_na * d omega_na / d z_nd,       dr[c][n] = sum_a gb[c][u_a] omega_na
// One wave per point: lane = (a' group, a); partial sums meet by xor-shuffles (fixed order: deterministic).
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_gram_bwd_kernel(const int* __restrict__ base, const T* __restrict__ w,
                                                           const T* __restrict__ dw, const T* __restrict__ r,
                                                           const T* __restrict__ GA, const T* __restrict__ gb, int N,
                                                           int G, int C, T* __restrict__ dZ, T* __restrict__ dr) {
    constexpr int S = DIM == 1 ? 4 : 16;          // stencil entries
    constexpr int GR = 64 / S;                    // lane groups: each takes S / GR of the a' entries
    constexpr int PER = S / GR;                   // (D = 1: 16 groups would exceed S -> groups beyond S idle, see below)
    const int m = DIM == 1 ? G : G * G;
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = n < N;
    const int nn = live ? n : 0;
    const int a = lane % S, grp = lane / S;
    const T wa = ski_omega<T, DIM>(w, nn, a);
    const int ua = ski_node<DIM>(base, nn, a, G);
    T g = T(0);
    if (DIM == 1) {
        // 4 entries x 16 groups: group g < 4 handles a' = g
        if (grp < S) g = T(2) * GA[(size_t)ua * m + ski_node<DIM>(base, nn, grp, G)] * ski_omega<T, DIM>(w, nn, grp);
    } else {
        #pragma unroll
        for (int k = 0; k < (PER > 0 ? PER : 1); ++k) {
            const int a2 = grp * PER + k;
            g += T(2) * GA[(size_t)ua * m + ski_node<DIM>(base, nn, a2, G)] * ski_omega<T, DIM>(w, nn, a2);
        }
    }
    // sum over the lane groups (lanes with equal a): xor offsets S, 2S, ... < 64
    for (int o = S; o < 64; o <<= 1) g += __shfl_xor(g, o);
    T drc[SKI_MAXC];
    #pragma unroll
    for (int c = 0; c < SKI_MAXC; ++c) drc[c] = T(0);
    for (int c = 0; c < C; ++c) {
        const T gbu = gb[(size_t)c * m + ua];
        g += gbu * r[(size_t)c * N + nn];
        drc[c] = gbu * wa;
    }
    // per-dimension derivative of omega_a, then the sum over the S entries (xor offsets 1 .. S/2)
    T dz0, dz1 = T(0);
    if (DIM == 1) dz0 = g * dw[(size_t)nn * 4 + a];
    else {
        dz0 = g * dw[((size_t)nn * 2) * 4 + (a >> 2)] * w[((size_t)nn * 2 + 1) * 4 + (a & 3)];
        dz1 = g * w[((size_t)nn * 2) * 4 + (a >> 2)] * dw[((size_t)nn * 2 + 1) * 4 + (a & 3)];
    }
    for (int o = 1; o < S; o <<= 1) {
        dz0 += __shfl_xor(dz0, o);
        if (DIM == 2) dz1 += __shfl_xor(dz1, o);
        #pragma unroll
        for (int c = 0; c < SKI_MAXC; ++c) if (c < C) drc[c] += __shfl_xor(drc[c], o);
    }
    if (live && lane == 0) {
        dZ[(size_t)n * DIM] = dz0;
        if (DIM == 2) dZ[(size_t)n * 2 + 1] = dz1;
        for (int c = 0; c < C; ++c) dr[(size_t)c * N + n] = drc[c];
    }
}

// ------------------------------------------------------------------ Y[c][n] = sum_a omega_na V[c][u_na]
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_interp_kernel(const int* __restrict__ base, const T* __restrict__ w,
                                                         const T* __restrict__ V, int N, int G, int C,
                                                         T* __restrict__ Y) {
    constexpr int S = DIM == 1 ? 4 : 16;
    const int m = DIM == 1 ? G : G * G;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * N) return;
    const int c = e / N, n = e - c * N;
    T acc = T(0);
    #pragma unroll
    for (int a = 0; a < S; ++a) acc += ski_omega<T, DIM>(w, n, a) * V[(size_t)c * m + ski_node<DIM>(base, n, a, G)];
    Y[e] = acc;
}

// ------------------------------------------------------------------ out[i][j] = scale * w1_i^T Q w2_j  (diag: j == i)
template <typename T, int DIM>
__global__ __launch_bounds__(256) void ski_cov_kernel(const int* __restrict__ base1, const T* __restrict__ w1, int N1,
                                                      const int* __restrict__ base2, const T* __restrict__ w2, int N2,
                                                      const T* __restrict__ Q, int G, T scale, int diag,
                                                      T* __restrict__ out) {
    constexpr int S = DIM == 1 ? 4 : 16;
    const int m = DIM == 1 ? G : G * G;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = diag ? (size_t)N1 : (size_t)N1 * N2;
    if (e >= tot) return;
    const int i = diag ? (int)e : (int)(e / N2), j = diag ? (int)e : (int)(e - (size_t)i * N2);
    int u2[S]; T o2[S];
    #pragma unroll
    for (int b = 0; b < S; ++b) { u2[b] = ski_node<DIM>(base2, j, b, G); o2[b] = ski_omega<T, DIM>(w2, j, b); }
    T acc = T(0);
    for (int a = 0; a < S; ++a) {
        const T* q = Q + (size_t)ski_node<DIM>(base1, i, a, G) * m;
        T row = T(0);
        #pragma unroll
        for (int b = 0; b < S; ++b) row += q[u2[b]] * o2[b];
        acc += ski_omega<T, DIM>(w1, i, a) * row;
    }
    out[e] = scale * acc;
}

// ================================================================== C ABI
template <typename T>
static int ski_weights_t(const void* Z, const void* g0, const void* inv_delta, int N, int D, int G, int* base, void* w,
                         void* dw, int* cell, hipStream_t st) {
    AMX_LAUNCH((ski_weights_kernel<T>), dim3(amx_ceil_div(N * D, 256)), dim3(256), 0, st, (const T*)Z, (const T*)g0,
               (const T*)inv_delta, N, D, G, base, (T*)w, (T*)dw, cell);
    AMX_CHECK_LAUNCH();
    return 0;
}
extern "C" int amx_ski_weights(const void* Z, const void* g0, const void* inv_delta, int N, int D, int G, int is_double,
                               int* base, void* w, void* dw, int* cell, void* stream) {
    if (!Z || !g0 || !inv_delta || !base || !w || !dw) AMX_BADARG(1);
    if (N <= 0 || (D != 1 && D != 2) || G < 4) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    return is_double ? ski_weights_t<double>(Z, g0, inv_delta, N, D, G, base, w, dw, cell, st)
                     : ski_weights_t<float>(Z, g0, inv_delta, N, D, G, base, w, dw, cell, st);
}

template <typename T, int DIM>
static int ski_gram_t(const void* w, const void* r, const int* order, const int* cell_start, int N, int G, int C, void* ws,
                      void* A, void* b, hipStream_t st) {
    constexpr int S = DIM == 1 ? 4 : 16;
    const int NC = G - 3, ncell = DIM == 1 ? NC : NC * NC, m = DIM == 1 ? G : G * G;
    T* blocks = (T*)ws;
    T* bvec = blocks + (size_t)ncell * S * S;
    AMX_LAUNCH((ski_cell_blocks_kernel<T, DIM>), dim3(ncell), dim3(256), 0, st, (const T*)w, (const T*)r, order, cell_start,
               N, C, blocks, bvec);
    AMX_CHECK_LAUNCH();
    if (A) {
        AMX_LAUNCH((ski_gram_gather_kernel<T, DIM>), dim3(amx_ceil_div(m * (DIM == 1 ? 7 : 49), 256)), dim3(256), 0, st,
                   (const T*)blocks, G, (T*)A);
        AMX_CHECK_LAUNCH();
    }
    if (C > 0) {
        AMX_LAUNCH((ski_bvec_gather_kernel<T, DIM>), dim3(amx_ceil_div(C * m, 256)), dim3(256), 0, st, (const T*)bvec, G, C,
                   (T*)b);
        AMX_CHECK_LAUNCH();
    }
    return 0;
}
extern "C" long amx_ski_gram_workspace(int D, int G, int C) {
    if ((D != 1 && D != 2) || G < 4 || C < 0 || C > SKI_MAXC) return -1;
    const long S = D == 1 ? 4 : 16, NC = G - 3, ncell = D == 1 ? NC : NC * NC;
    return ncell * (S * S + (long)C * S);                       // elements of the value type
}
extern "C" int amx_ski_gram(const void* w, const void* r, const int* order, const int* cell_start, int N, int D, int G,
                            int C, int is_double, void* ws, void* A, void* b, void* stream) {
    if (!w || !order || !cell_start || !ws || (C > 0 && (!r || !b))) AMX_BADARG(1);
    if (N <= 0 || (D != 1 && D != 2) || G < 4 || C < 0 || C > SKI_MAXC || (!A && C == 0)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    if (is_double) return D == 1 ? ski_gram_t<double, 1>(w, r, order, cell_start, N, G, C, ws, A, b, st)
                                 : ski_gram_t<double, 2>(w, r, order, cell_start, N, G, C, ws, A, b, st);
    return D == 1 ? ski_gram_t<float, 1>(w, r, order, cell_start, N, G, C, ws, A, b, st)
                  : ski_gram_t<float, 2>(w, r, order, cell_start, N, G, C, ws, A, b, st);
}

template <typename T, int DIM>
static int ski_gram_bwd_t(const int* base, const void* w, const void* dw, const void* r, const void* GA, const void* gb,
                          int N, int G, int C, void* dZ, void* dr, hipStream_t st) {
    AMX_LAUNCH((ski_gram_bwd_kernel<T, DIM>), dim3(amx_ceil_div(N, 4)), dim3(256), 0, st, base, (const T*)w, (const T*)dw,
               (const T*)r, (const T*)GA, (const T*)gb, N, G, C, (T*)dZ, (T*)dr);
    AMX_CHECK_LAUNCH();
    return 0;
}
extern "C" int amx_ski_gram_bwd(const int* base, const void* w, const void* dw, const void* r, const void* GA,
                                const void* gb, int N, int D, int G, int C, int is_double, void* dZ, void* dr,
                                void* stream) {
    if (!base || !w || !dw || !GA || !dZ || (C > 0 && (!r || !gb || !dr))) AMX_BADARG(1);
    if (N <= 0 || (D != 1 && D != 2) || G < 4 || C < 0 || C > SKI_MAXC) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    if (is_double) return D == 1 ? ski_gram_bwd_t<double, 1>(base, w, dw, r, GA, gb, N, G, C, dZ, dr, st)
                                 : ski_gram_bwd_t<double, 2>(base, w, dw, r, GA, gb, N, G, C, dZ, dr, st);
    return D == 1 ? ski_gram_bwd_t<float, 1>(base, w, dw, r, GA, gb, N, G, C, dZ, dr, st)
                  : ski_gram_bwd_t<float, 2>(base, w, dw, r, GA, gb, N, G, C, dZ, dr, st);
}

template <typename T, int DIM>
static int ski_interp_t(const int* base, const void* w, const void* V, int N, int G, int C, void* Y, hipStream_t st) {
    AMX_LAUNCH((ski_interp_kernel<T, DIM>), dim3(amx_ceil_div(C * N, 256)), dim3(256), 0, st, base, (const T*)w,
               (const T*)V, N, G, C, (T*)Y);
    AMX_CHECK_LAUNCH();
    return 0;
}
extern "C" int amx_ski_interp(const int* base, const void* w, const void* V, int N, int D, int G, int C, int is_double,
                              void* Y, void* stream) {
    if (!base || !w || !V || !Y) AMX_BADARG(1);
    if (N <= 0 || (D != 1 && D != 2) || G < 4 || C <= 0) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    if (is_double) return D == 1 ? ski_interp_t<double, 1>(base, w, V, N, G, C, Y, st) : ski_interp_t<double, 2>(base, w, V, N, G, C, Y, st);
    return D == 1 ? ski_interp_t<float, 1>(base, w, V, N, G, C, Y, st) : ski_interp_t<float, 2>(base, w, V, N, G, C, Y, st);
}

template <typename T, int DIM>
static int ski_cov_t(const int* base1, const void* w1, int N1, const int* base2, const void* w2, int N2, const void* Q, int G,
                     double scale, int diag, void* out, hipStream_t st) {
    const size_t tot = diag ? (size_t)N1 : (size_t)N1 * N2;
    AMX_LAUNCH((ski_cov_kernel<T, DIM>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, base1, (const T*)w1, N1,
               base2, (const T*)w2, N2, (const T*)Q, G, (T)scale, diag, (T*)out);
    AMX_CHECK_LAUNCH();
    return 0;
}
extern "C" int amx_ski_cov(const int* base1, const void* w1, int N1, const int* base2, const void* w2, int N2,
                           const void* Q, int D, int G, double scale, int diag, int is_double, void* out, void* stream) {
    if (!base1 || !w1 || !base2 || !w2 || !Q || !out) AMX_BADARG(1);
    if (N1 <= 0 || N2 <= 0 || (D != 1 && D != 2) || G < 4 || (diag && N1 != N2)) AMX_BADARG(2);
    hipStream_t st = (hipStream_t)stream;
    if (is_double) return D == 1 ? ski_cov_t<double, 1>(base1, w1, N1, base2, w2, N2, Q, G, scale, diag, out, st)
                                 : ski_cov_t<double, 2>(base1, w1, N1, base2, w2, N2, Q, G, scale, diag, out, st);
    return D == 1 ? ski_cov_t<float, 1>(base1, w1, N1, base2, w2, N2, Q, G, scale, diag, out, st)
                  : ski_cov_t<float, 2>(base1, w1, N1, base2, w2, N2, Q, G, scale, diag, out, st);
}
