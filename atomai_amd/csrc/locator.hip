// locator.hip — pixel-wise class probabilities -> blob centres, on the device.
//
// Replaces the per-frame CPU loop of the reference's Locator (atomai/predictors/predictor.py:584-611):
//   cv_thresh  (atomai/utils/img.py:554-564)   binary threshold  x > t -> 1.0 else 0
//   find_com   (atomai/utils/coords.py:21-34)   scipy.ndimage.label (4-connectivity) + center_of_mass of the
//                                              binary image, one (row, col) per label, in label order
//   rem_edge_coord (predictor.py:621-639)       drop centres closer than dist_edge to the frame border
// for every frame and every class channel except the last (background).
//
// Integer/byte work bound by HBM: no MFMA.  One linear element space e = ((b*nch + c)*H + h)*W + w.
//   1. locate_init     L[e] = e for foreground pixels, -1 otherwise (reads the NHWC probabilities once)
//   2. locate_merge    lock-free union-find (atomicMin on roots) over left / up neighbours
//   3. locate_flatten  L[e] = root(e); per-root pixel count and row / column sums (integer atomics — exact,
//                      order independent, so results are bit-reproducible)
//   4. locate_count / locate_scan / locate_emit  order-preserving stream compaction of the surviving roots.
// The root of a component is its smallest linear index = its first pixel in raster order, which is exactly
// the order in which scipy.ndimage.label numbers components; centres are sums/count in fp64, the arithmetic
// center_of_mass performs (integer-valued sums are exact in both).
#include "amx_device.h"

typedef unsigned long long u64;

#define LOC_STRIP 8                    // consecutive elements per thread
#define LOC_CHUNK (256 * LOC_STRIP)    // elements per workgroup in the compaction passes

struct LocWork {                       // views into the caller's workspace
    int* L;
    unsigned* cnt;
    u64* sr;
    u64* sc;
    int* chunk_off;                    // [nchunks + 1]
};

static __host__ __device__ inline long loc_nchunks(long ne) { return (ne + LOC_CHUNK - 1) / LOC_CHUNK; }

static __host__ inline LocWork loc_views(void* work, long ne) {
    char* p = (char*)work;
    LocWork w;
    w.sr = (u64*)p;            p += ne * 8;
    w.sc = (u64*)p;            p += ne * 8;
    w.L = (int*)p;             p += ne * 4;
    w.cnt = (unsigned*)p;      p += ne * 4;
    w.chunk_off = (int*)p;
    return w;
}

static __device__ __forceinline__ int loc_load(const int* p) {
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}

static __device__ __forceinline__ int loc_find(const int* L, int i) {
    int p;
    while ((p = loc_load(L + i)) != i) i = p;
    return i;
}

static __device__ __forceinline__ void loc_unite(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = loc_find(L, a);
        b = loc_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + b, a);         // hook the larger root under the smaller one
        done = (old == b);
        b = old;
    }
}

__global__ __launch_bounds__(256) void locate_init_kernel(const float* __restrict__ prob, int* __restrict__ L,
                                                          unsigned* __restrict__ cnt, u64* __restrict__ sr,
                                                          u64* __restrict__ sc, long ne, int H, int W, int C,
                                                          int nch, float thr) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int w = (int)(e % W);
    long t = e / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % nch);
    const long b = t / nch;
    const float p = prob[((b * H + h) * (long)W + w) * C + c];
    const bool fg = p > thr;                        // cv2.THRESH_BINARY: strictly greater; NaN -> background
    L[e] = fg ? (int)e : -1;
    if (fg) { cnt[e] = 0u; sr[e] = 0ull; sc[e] = 0ull; }
}

__global__ __launch_bounds__(256) void locate_merge_kernel(int* __restrict__ L, long ne, int H, int W) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    if (loc_load(L + e) < 0) return;
    const int w = (int)(e % W);
    const int h = (int)((e / W) % H);
    const bool left = w > 0 && loc_load(L + e - 1) >= 0;
    const bool up = h > 0 && loc_load(L + e - W) >= 0;
    if (left) loc_unite(L, (int)e, (int)e - 1);
    // up is already connected through left + up-left when all three are foreground
    if (up && !(left && loc_load(L + e - W - 1) >= 0)) loc_unite(L, (int)e, (int)e - W);
}

__global__ __launch_bounds__(256) void locate_flatten_kernel(int* __restrict__ L, unsigned* __restrict__ cnt,
                                                             u64* __restrict__ sr, u64* __restrict__ sc,
                                                             long ne, int H, int W) {
    const long e0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * LOC_STRIP;
    if (e0 >= ne) return;
    int root = -1;
    unsigned n = 0;
    u64 ar = 0, ac = 0;
    for (int j = 0; j < LOC_STRIP; ++j) {
        const long e = e0 + j;
        if (e >= ne) break;
        int r = -1;
        if (loc_load(L + e) >= 0) {
            r = loc_find(L, (int)e);
            L[e] = r;                                // every ancestor is valid for concurrent readers
        }
        if (r != root) {                             // flush the run accumulated so far
            if (root >= 0) { atomicAdd(cnt + root, n); atomicAdd(sr + root, ar); atomicAdd(sc + root, ac); }
            root = r; n = 0; ar = 0; ac = 0;
        }
        if (r >= 0) { n += 1u; ar += (u64)((e / W) % H); ac += (u64)(e % W); }
    }
    if (root >= 0) { atomicAdd(cnt + root, n); atomicAdd(sr + root, ar); atomicAdd(sc + root, ac); }
}

// A root survives when its centre is not within dist_edge of the border (predictor.py:625-633, fp64 compares).
static __device__ __forceinline__ bool loc_keep(const int* L, const unsigned* cnt, const u64* sr, const u64* sc,
                                                long e, int H, int W, int dist_edge, double* row, double* col) {
    if (L[e] != (int)e) return false;
    const double n = (double)cnt[e];
    const double r = (double)sr[e] / n, c = (double)sc[e] / n;
    *row = r; *col = c;
    return !(r > (double)(H - dist_edge) || r < (double)dist_edge || c > (double)(W - dist_edge) ||
             c < (double)dist_edge);
}

__global__ __launch_bounds__(256) void locate_count_kernel(const int* __restrict__ L, const unsigned* __restrict__ cnt,
                                                           const u64* __restrict__ sr, const u64* __restrict__ sc,
                                                           int* __restrict__ chunk_cnt, long ne, int H, int W,
                                                           int dist_edge) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const long e0 = (long)blockIdx.x * LOC_CHUNK + (long)threadIdx.x * LOC_STRIP;
    int mine = 0;
    double r, c;
    for (int j = 0; j < LOC_STRIP; ++j)
        if (e0 + j < ne && loc_keep(L, cnt, sr, sc, e0 + j, H, W, dist_edge, &r, &c)) ++mine;
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = total;
}

// Exclusive scan of the per-chunk counts in place (single workgroup); chunk_off[nchunks] and *count = total.
__global__ __launch_bounds__(1024) void locate_scan_kernel(int* __restrict__ chunk_off, long nchunks,
                                                           int* __restrict__ count) {
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long base = 0; base < nchunks; base += 1024) {
        const long i = base + threadIdx.x;
        const int v = i < nchunks ? chunk_off[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan
            const int add = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = buf[threadIdx.x], c0 = carry;
        if (i < nchunks) chunk_off[i] = c0 + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { chunk_off[nchunks] = carry; *count = carry; }
}

__global__ __launch_bounds__(256) void locate_emit_kernel(const int* __restrict__ L, const unsigned* __restrict__ cnt,
                                                          const u64* __restrict__ sr, const u64* __restrict__ sc,
                                                          const int* __restrict__ chunk_off, double* __restrict__ coords,
                                                          int* __restrict__ meta, long cap, long ne, int H, int W,
                                                          int nch, int dist_edge) {
    __shared__ int pre[256];
    const long e0 = (long)blockIdx.x * LOC_CHUNK + (long)threadIdx.x * LOC_STRIP;
    double rr[LOC_STRIP], cc[LOC_STRIP];
    unsigned mask = 0;
    int mine = 0;
    for (int j = 0; j < LOC_STRIP; ++j)
        if (e0 + j < ne && loc_keep(L, cnt, sr, sc, e0 + j, H, W, dist_edge, &rr[j], &cc[j])) { mask |= 1u << j; ++mine; }
    pre[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int add = (int)threadIdx.x >= d ? pre[threadIdx.x - d] : 0;
        __syncthreads();
        pre[threadIdx.x] += add;
        __syncthreads();
    }
    long o = (long)chunk_off[blockIdx.x] + pre[threadIdx.x] - mine;
    for (int j = 0; j < LOC_STRIP; ++j) {
        if (!(mask & (1u << j))) continue;
        if (o < cap) {
            const long e = e0 + j;
            const long fc = e / ((long)H * W);            // = b*nch + c
            coords[2 * o] = rr[j];
            coords[2 * o + 1] = cc[j];
            meta[2 * o] = (int)(fc / nch);
            meta[2 * o + 1] = (int)(fc % nch);
        }
        ++o;
    }
}

// ---------------------------------------------------------------------------------------------- C ABI
extern "C" long amx_locate_workspace_bytes(int B, int H, int W, int nch) {
    if (B <= 0 || H <= 0 || W <= 0 || nch <= 0) AMX_BADARG(1);
    const long ne = (long)B * nch * H * W;
    if (ne >= 2147483647L) AMX_BADARG(2);                // int32 labels: chunk the stack on the host
    return ne * 24 + (loc_nchunks(ne) + 1) * 4 + 64;
}

extern "C" int amx_locate_label(const float* prob, int B, int H, int W, int C, int nch, float thr, int dist_edge,
                                void* work, int* count, void* stream) {
    if (!prob || !work || !count) AMX_BADARG(1);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || nch <= 0 || nch > C) AMX_BADARG(2);
    const long ne = (long)B * nch * H * W;
    if (ne >= 2147483647L) AMX_BADARG(3);
    if ((uintptr_t)work & 7) AMX_BADARG(4);
    const LocWork w = loc_views(work, ne);
    hipStream_t s = (hipStream_t)stream;
    const unsigned g1 = (unsigned)((ne + 255) / 256);
    AMX_LAUNCH(locate_init_kernel, dim3(g1), dim3(256), 0, s, prob, w.L, w.cnt, w.sr, w.sc, ne, H, W, C, nch, thr);
    AMX_LAUNCH(locate_merge_kernel, dim3(g1), dim3(256), 0, s, w.L, ne, H, W);
    const long nchunks = loc_nchunks(ne);
    AMX_LAUNCH(locate_flatten_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, w.L, w.cnt, w.sr, w.sc, ne, H, W);
    AMX_LAUNCH(locate_count_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, (const int*)w.L,
               (const unsigned*)w.cnt, (const u64*)w.sr, (const u64*)w.sc, w.chunk_off, ne, H, W, dist_edge);
    AMX_LAUNCH(locate_scan_kernel, dim3(1), dim3(1024), 0, s, w.chunk_off, nchunks, count);
    AMX_CHECK_LAUNCH();
    return 0;
}

extern "C" int amx_locate_emit(const void* work, int B, int H, int W, int nch, int dist_edge, double* coords,
                               int* meta, long cap, void* stream) {
    if (!work || B <= 0 || H <= 0 || W <= 0 || nch <= 0) AMX_BADARG(1);
    if (cap < 0 || (cap > 0 && (!coords || !meta))) AMX_BADARG(2);
    const long ne = (long)B * nch * H * W;
    if (ne >= 2147483647L) AMX_BADARG(3);
    if (cap == 0) return 0;
    const LocWork w = loc_views(const_cast<void*>(work), ne);
    AMX_LAUNCH(locate_emit_kernel, dim3((unsigned)loc_nchunks(ne)), dim3(256), 0, (hipStream_t)stream,
               (const int*)w.L, (const unsigned*)w.cnt, (const u64*)w.sr, (const u64*)w.sc,
               (const int*)w.chunk_off, coords, meta, cap, ne, H, W, nch, dist_edge);
    AMX_CHECK_LAUNCH();
    return 0;
}
