// locator.hip — pixel-wise class probabilities -> blob centres, on the device.
//
// Replaces the per-frame CPU loop of the reference's Locator (atomai/predictors/predictor.py:584-611):
//   cv_thresh  (atomai/utils/img.py:554-564)   binary threshold  x > t -> 1.0 else 0
//   find_com   (atomai/utils/coords.py:21-34)   scipy.ndimage.label (4-connectivity) + center_of_mass of the
//                                              binary image, one (row, col) per label, in label order
//   rem_edge_coord (predictor.py:621-639)       drop centres closer than dist_edge to the frame border
// for every frame and every class channel except the last (background).
//
// Integer/byte work bound by HBM: no MFMA.  One linear element space e = ((b*nch + c)*H + h)*W + w (a "plane" is
// one (frame, class) map).  Round 3 — labelling happens in LDS, global memory only sees what crosses a tile:
//   1. locate_tile     one workgroup per 32 x 64-pixel tile: reads the NHWC probabilities ONCE (8 consecutive pixels
//                      per thread), thresholds, labels the tile's 4-connected components with a union-find over
//                      horizontal RUNS in LDS (atomicMin on LDS parents; a run never needs more than one hook per
//                      overlapping run of the row above), sums count / row / column per component with one packed
//                      64-bit LDS atomic per run, and writes ONLY what later passes read: the root id (element index
//                      of the tile-local root, -1: background) of the pixels on the tile's four sides into compact
//                      border arrays, parent = self and the sums at the local roots, and one BYTE of "is a local
//                      root" bits per 8 pixels.  Global traffic: 4 B read + ~0.6 B written per pixel (a first version
//                      wrote the root id of every pixel: 79 us instead of 60 for 32 frames of 1024^2);
//   2. locate_border   lock-free union-find (atomicMin on roots) over the pixel pairs that straddle a tile border —
//                      1/32 + 1/64 of the pixels, read from the border arrays (coalesced), chains start at local roots;
//   3. locate_fold     every local root that was hooked under another one adds its sums to its final root
//                      (integer atomics — exact, order independent, so results are bit-reproducible);
//   4. locate_count / locate_scan / locate_emit  order-preserving stream compaction of the surviving roots; these
//                      passes walk the root BIT MAP (1/32 of the label volume), not the labels.
// The root of a component is its smallest linear index = its first pixel in raster order, which is exactly
// the order in which scipy.ndimage.label numbers components; centres are sums/count in fp64, the arithmetic
// center_of_mass performs (integer-valued sums are exact in both).
// (Round 1-2 labelled with ONE global union-find over all pixels: 0.65 ms per 32 frames of 1024^2 = 2.6 % of the HBM
// roofline, parent chains chased through L2; now 0.121 ms = 13.8 %.  profiles/r03_locator_tiles.md has the steps,
// including the persistent / prefetching variant of pass 1 that was SLOWER: the pass is bound by its LDS phases and
// barriers, not by the latency of its one global load.)
#include "amx_device.h"

typedef unsigned long long u64;

#define LT_H 32                        // tile rows
#define LT_W 64                        // tile columns (8 strips of 8 pixels: one thread each)
#define LT_N (LT_H * LT_W)
#define LOC_MB 8                       // root-map bytes (= 64 pixels) per thread in the fold / compaction passes
#define LOC_CHUNK (256 * LOC_MB)       // root-map bytes per workgroup in the compaction passes

struct LocWork {                       // views into the caller's workspace
    int* L;                            // [ne]   parent pointers, valid at tile-local roots only
    unsigned* cnt;                     // [ne]   valid at local roots only
    u64* sr;                           // [ne]   "
    u64* sc;                           // [ne]   "
    unsigned char* rmap;               // [planes * H][WB] one bit per pixel: tile-local root
    int* chunk_off;                    // [nchunks + 1]
    int* rowB;                         // [planes][nty][2][W]  root ids of the first / last row of every tile row
    int* colB;                         // [planes][ntx][2][H]  root ids of the first / last column of every tile column
};

static __host__ __device__ inline long loc_wb(int W) { return (W + 7) / 8; }
static __host__ __device__ inline long loc_map_bytes(long rows, int W) { return (rows * loc_wb(W) + 7) / 8 * 8; }
static __host__ __device__ inline long loc_nchunks(long map_bytes) { return (map_bytes + LOC_CHUNK - 1) / LOC_CHUNK; }

static __host__ inline long loc_border_ints(long planes, int H, int W) {
    return planes * 2 * ((long)amx_ceil_div(H, LT_H) * W + (long)amx_ceil_div(W, LT_W) * H);
}

static __host__ inline LocWork loc_views(void* work, long ne, long map_bytes, long planes, int H, int W) {
    char* p = (char*)work;
    LocWork w;
    w.sr = (u64*)p;            p += ne * 8;
    w.sc = (u64*)p;            p += ne * 8;
    w.L = (int*)p;             p += ne * 4;
    w.cnt = (unsigned*)p;      p += ne * 4;
    p += (8 - ((uintptr_t)p & 7)) & 7;
    w.rmap = (unsigned char*)p; p += map_bytes;
    w.chunk_off = (int*)p;     p += (loc_nchunks(map_bytes) + 1) * 4;
    p += (16 - ((uintptr_t)p & 15)) & 15;
    w.rowB = (int*)p;          p += planes * amx_ceil_div(H, LT_H) * 2 * (long)W * 4;
    w.colB = (int*)p;
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

static __device__ __forceinline__ int loc_ctz(unsigned v) {
#ifdef AMX_EMU
    return __builtin_ctz(v);
#else
    return __ffs((int)v) - 1;
#endif
}
static __device__ __forceinline__ int loc_popc(unsigned v) {
#ifdef AMX_EMU
    return __builtin_popcount(v);
#else
    return __popc(v);
#endif
}
// Sum slot of the run that starts at local pixel r: a strip of 8 pixels holds at most 4 runs
static __device__ __forceinline__ int loc_slot(const u64* s_row, int r) {
    const unsigned m = reinterpret_cast<const unsigned char*>(s_row)[r >> 3];
    return (r >> 3) * 4 + loc_popc(m & ~(m << 1) & ((1u << (r & 7)) - 1u));
}

// Pass 1.  Thread t owns the strip of 8 pixels at tile row t / 8, columns 8 * (t % 8) ...: local index t * 8 + j.
// The 8 strips of a tile row sit in 8 consecutive lanes of one wave, so the 64-bit foreground mask R of the row is
// available after a wave-level exchange and the horizontal runs need no union-find at all: the run that contains
// column x starts right above the highest zero bit of R below x.  Only vertical overlaps are hooked (once per overlap
// run of R & R_above), and the parent chains start at row-run level.
static __device__ __forceinline__ int loc_run_start(u64 R, int x) {           // bit x of R is set
    const u64 below = ~R & ((1ull << x) - 1ull);
    return below ? 64 - __builtin_clzll(below) : 0;
}

__global__ __launch_bounds__(256) void locate_tile_kernel(const float* __restrict__ prob, int* __restrict__ L,
                                                          unsigned* __restrict__ cnt, u64* __restrict__ sr,
                                                          u64* __restrict__ sc, unsigned char* __restrict__ rmap,
                                                          int* __restrict__ rowB, int* __restrict__ colB,
                                                          int H, int W, int C, int nch, float thr, int ntx, int nty) {
    __shared__ int s_lab[LT_N];                      // parent (local index); -1 background
    __shared__ u64 s_acc[LT_N / 2];                  // [strip][run]: count | row sum << 16 | column sum << 40 (tile-local)
    __shared__ u64 s_row[LT_H];                      // foreground mask of every tile row (byte scol = strip scol)
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty;
    const int fc = t / nty;                          // plane = b * nch + c
    const int srow = tid >> 3, scol = tid & 7;
    const int h = ty * LT_H + srow, w0 = tx * LT_W + scol * 8;
    const int base = tid * 8, rbase = srow * LT_W;
    const int nvalid = h >= H ? 0 : (W - w0 >= 8 ? 8 : (W - w0 > 0 ? W - w0 : 0));

    unsigned m = 0;
    if (nvalid) {
        const long pix = ((long)(fc / nch) * H + h) * W + w0;
        const float* src = prob + pix * C + (fc % nch);
        if (C == 1 && nvalid == 8 && (pix & 3) == 0) {
            const float4 a = amx_ld4(src), b = amx_ld4(src + 4);
            m = (a.x > thr ? 1u : 0u) | (a.y > thr ? 2u : 0u) | (a.z > thr ? 4u : 0u) | (a.w > thr ? 8u : 0u) |
                (b.x > thr ? 16u : 0u) | (b.y > thr ? 32u : 0u) | (b.z > thr ? 64u : 0u) | (b.w > thr ? 128u : 0u);
        } else {
            for (int j = 0; j < nvalid; ++j)
                if (src[(long)j * C] > thr) m |= 1u << j;    // cv2.THRESH_BINARY: strictly greater; NaN -> background
        }
    }
    reinterpret_cast<unsigned char*>(s_row)[tid] = (unsigned char)m;
    s_acc[tid * 4] = 0ull; s_acc[tid * 4 + 1] = 0ull; s_acc[tid * 4 + 2] = 0ull; s_acc[tid * 4 + 3] = 0ull;
    amx_wave_sync();                                 // the 8 strips of a row are lanes of one wave
    const u64 R = s_row[srow];
    {
        int cur = -1;
        #pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (m & (1u << j)) {
                if (cur < 0) cur = rbase + loc_run_start(R, scol * 8 + j);
            } else cur = -1;
            s_lab[base + j] = cur;                   // every pixel of a row run points at the run's first pixel
        }
    }
    __syncthreads();
    if (m && srow > 0) {
        const u64 Rup = s_row[srow - 1];
        const u64 both = R & Rup;
        unsigned starts = (unsigned)((both & ~(both << 1)) >> (scol * 8)) & 0xFFu;   // one hook per overlap run
        while (starts) {
            const int x = scol * 8 + loc_ctz(starts);
            starts &= starts - 1;
            loc_unite(s_lab, rbase + loc_run_start(R, x), rbase - LT_W + loc_run_start(Rup, x));
        }
    }
    __syncthreads();
    // flatten: the labels are read-only from here on; one LDS atomic per run
    int lab[8];
    {
        int r = -1, n = 0, cs = 0;
        #pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (m & (1u << j)) {
                if (n == 0) r = loc_find(s_lab, s_lab[base + j]);
                ++n; cs += scol * 8 + j;
                lab[j] = r;
            } else {
                if (n) atomicAdd(&s_acc[loc_slot(s_row, r)], (u64)n | ((u64)(n * srow) << 16) | ((u64)cs << 40));
                n = 0; cs = 0;
                lab[j] = -1;
            }
        }
        if (n) atomicAdd(&s_acc[loc_slot(s_row, r)], (u64)n | ((u64)(n * srow) << 16) | ((u64)cs << 40));
    }
    const long tile0 = (long)fc * H * W + (long)(ty * LT_H) * W + tx * LT_W;   // element of the tile's first pixel
    if (nvalid) {
        // root ids (element index of the tile-local root) of the pixels on the tile's sides, for pass 2
        const bool first_row = srow == 0 && ty > 0, last_row = srow == LT_H - 1 && ty + 1 < nty;
        if (first_row || last_row) {
            int* dst = rowB + (((long)fc * nty + ty) * 2 + (last_row ? 1 : 0)) * W + w0;
            #pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nvalid) dst[j] = lab[j] < 0 ? -1 : (int)(tile0 + (long)(lab[j] >> 6) * W + (lab[j] & 63));
        }
        if (scol == 0 && tx > 0)
            colB[(((long)fc * ntx + tx) * 2) * H + h] = lab[0] < 0 ? -1 : (int)(tile0 + (long)(lab[0] >> 6) * W + (lab[0] & 63));
        if (scol == 7 && tx + 1 < ntx)               // (a tile that has a right neighbour is full width)
            colB[(((long)fc * ntx + tx) * 2 + 1) * H + h] = lab[7] < 0 ? -1 : (int)(tile0 + (long)(lab[7] >> 6) * W + (lab[7] & 63));
    }
    __syncthreads();                                 // the sums are complete
    if (nvalid) {
        unsigned roots = 0;
        unsigned cand = m & ~(m << 1);               // strip-run starts: the only pixels that can be roots
        for (int slot = tid * 4; cand; ++slot) {
            const int j = loc_ctz(cand);
            cand &= cand - 1;
            if (s_lab[base + j] != base + j) continue;
            roots |= 1u << j;
            const u64 a = s_acc[slot];
            const u64 n = a & 0xFFFFull;
            const long e = tile0 + (long)srow * W + scol * 8 + j;
            L[e] = (int)e;
            cnt[e] = (unsigned)n;
            sr[e] = ((a >> 16) & 0xFFFFFFull) + n * (u64)(ty * LT_H);
            sc[e] = (a >> 40) + n * (u64)(tx * LT_W);
        }
        rmap[((long)fc * H + h) * loc_wb(W) + (w0 >> 3)] = (unsigned char)roots;
    }
}

// Pass 2.  Item i of a plane: the (nty - 1) * W pixel pairs across the horizontal tile borders, then the (ntx - 1) * H
// pairs across the vertical ones; both sides are read from the border arrays (root ids), so the chains start at roots.
__global__ __launch_bounds__(256) void locate_border_kernel(int* __restrict__ L, const int* __restrict__ rowB,
                                                            const int* __restrict__ colB, int H, int W, int ntx, int nty,
                                                            long per_plane, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long fc = i / per_plane;
    long k = i - fc * per_plane;
    const long nrow_items = (long)(nty - 1) * W;
    if (k < nrow_items) {
        const int t = (int)(k / W) + 1, w = (int)(k % W);
        const int* below = rowB + ((fc * nty + t) * 2) * W;          // first row of tile row t
        const int* above = rowB + ((fc * nty + t - 1) * 2 + 1) * W;  // last row of tile row t - 1
        const int a = below[w], b = above[w];
        if (a < 0 || b < 0) return;
        // already connected through left + up-left when all four are foreground (that pair is handled by the item of
        // the pixel to the left, by the column items, or inside the tiles)
        if (w > 0 && below[w - 1] >= 0 && above[w - 1] >= 0) return;
        loc_unite(L, a, b);
    } else {
        k -= nrow_items;
        const int t = (int)(k / H) + 1, h = (int)(k % H);
        const int a = colB[((fc * ntx + t) * 2) * H + h], b = colB[((fc * ntx + t - 1) * 2 + 1) * H + h];
        if (a < 0 || b < 0) return;
        loc_unite(L, a, b);
    }
}

// Element index of bit `bit` of root-map byte `mb`.
static __device__ __forceinline__ long loc_elem(long mb, int bit, int W) {
    const long wb = loc_wb(W);
    const long row = mb / wb;
    return row * W + (mb - row * wb) * 8 + bit;
}

// Pass 3.  A tile-local root that is no longer a root hands its sums to the final root of its component.
__global__ __launch_bounds__(256) void locate_fold_kernel(const int* __restrict__ L, unsigned* __restrict__ cnt,
                                                          u64* __restrict__ sr, u64* __restrict__ sc,
                                                          const unsigned char* __restrict__ rmap, long map_bytes,
                                                          long live_bytes, int W) {
    const long b0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * LOC_MB;
    if (b0 >= map_bytes) return;
    u64 bits = *reinterpret_cast<const u64*>(rmap + b0);
    if (b0 + LOC_MB > live_bytes) bits &= b0 >= live_bytes ? 0ull : ((1ull << (8 * (live_bytes - b0))) - 1ull);
    while (bits) {
        const int q = __builtin_ctzll(bits);
        bits &= bits - 1;
        const long e = loc_elem(b0 + (q >> 3), q & 7, W);
        const int p = L[e];
        if (p == (int)e) continue;
        const int g = loc_find(L, p);
        atomicAdd(cnt + g, cnt[e]);
        atomicAdd(sr + g, sr[e]);
        atomicAdd(sc + g, sc[e]);
    }
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

// The 64 root-map bits of thread `threadIdx.x` of chunk `blockIdx.x` (bytes past the live map read as zero).
static __device__ __forceinline__ u64 loc_chunk_bits(const unsigned char* rmap, long map_bytes, long live_bytes, long* b0_out) {
    const long b0 = (long)blockIdx.x * LOC_CHUNK + (long)threadIdx.x * LOC_MB;
    *b0_out = b0;
    if (b0 >= map_bytes) return 0ull;
    u64 bits = *reinterpret_cast<const u64*>(rmap + b0);
    if (b0 + LOC_MB > live_bytes) bits &= b0 >= live_bytes ? 0ull : ((1ull << (8 * (live_bytes - b0))) - 1ull);
    return bits;
}

__global__ __launch_bounds__(256) void locate_count_kernel(const int* __restrict__ L, const unsigned* __restrict__ cnt,
                                                           const u64* __restrict__ sr, const u64* __restrict__ sc,
                                                           const unsigned char* __restrict__ rmap, long map_bytes,
                                                           long live_bytes, int* __restrict__ chunk_cnt, int H, int W,
                                                           int dist_edge) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    long b0;
    u64 bits = loc_chunk_bits(rmap, map_bytes, live_bytes, &b0);
    int mine = 0;
    double r, c;
    while (bits) {
        const int q = __builtin_ctzll(bits);
        bits &= bits - 1;
        if (loc_keep(L, cnt, sr, sc, loc_elem(b0 + (q >> 3), q & 7, W), H, W, dist_edge, &r, &c)) ++mine;
    }
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
                                                          const unsigned char* __restrict__ rmap, long map_bytes,
                                                          long live_bytes, const int* __restrict__ chunk_off,
                                                          double* __restrict__ coords, int* __restrict__ meta, long cap,
                                                          int H, int W, int nch, int dist_edge) {
    __shared__ int pre[256];
    long b0;
    const u64 all = loc_chunk_bits(rmap, map_bytes, live_bytes, &b0);
    u64 keep = 0;
    int mine = 0;
    double r, c;
    for (u64 bits = all; bits;) {
        const int q = __builtin_ctzll(bits);
        bits &= bits - 1;
        if (loc_keep(L, cnt, sr, sc, loc_elem(b0 + (q >> 3), q & 7, W), H, W, dist_edge, &r, &c)) { keep |= 1ull << q; ++mine; }
    }
    pre[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int add = (int)threadIdx.x >= d ? pre[threadIdx.x - d] : 0;
        __syncthreads();
        pre[threadIdx.x] += add;
        __syncthreads();
    }
    long o = (long)chunk_off[blockIdx.x] + pre[threadIdx.x] - mine;
    while (keep) {
        const int q = __builtin_ctzll(keep);
        keep &= keep - 1;
        if (o < cap) {
            const long e = loc_elem(b0 + (q >> 3), q & 7, W);
            loc_keep(L, cnt, sr, sc, e, H, W, dist_edge, &r, &c);
            const long fc = e / ((long)H * W);            // = b*nch + c
            coords[2 * o] = r;
            coords[2 * o + 1] = c;
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
    const long mb = loc_map_bytes((long)B * nch * H, W);
    return ne * 24 + 8 + mb + (loc_nchunks(mb) + 1) * 4 + 16 + loc_border_ints((long)B * nch, H, W) * 4 + 64;
}

extern "C" int amx_locate_label(const float* prob, int B, int H, int W, int C, int nch, float thr, int dist_edge,
                                void* work, int* count, void* stream) {
    if (!prob || !work || !count) AMX_BADARG(1);
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || nch <= 0 || nch > C) AMX_BADARG(2);
    const long ne = (long)B * nch * H * W;
    if (ne >= 2147483647L) AMX_BADARG(3);
    if ((uintptr_t)work & 7) AMX_BADARG(4);
    const long planes = (long)B * nch;
    const long live = planes * H * loc_wb(W), mb = loc_map_bytes(planes * H, W);
    const LocWork w = loc_views(work, ne, mb, planes, H, W);
    hipStream_t s = (hipStream_t)stream;
    const int ntx = amx_ceil_div(W, LT_W), nty = amx_ceil_div(H, LT_H);
    const long tiles = planes * ntx * nty;
    if (tiles >= 2147483647L) AMX_BADARG(5);
    AMX_LAUNCH(locate_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, s, prob, w.L, w.cnt, w.sr, w.sc, w.rmap, w.rowB,
               w.colB, H, W, C, nch, thr, ntx, nty);
    const long per_plane = (long)(nty - 1) * W + (long)(ntx - 1) * H;
    if (per_plane > 0) {
        const long total = per_plane * planes;
        AMX_LAUNCH(locate_border_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w.L, (const int*)w.rowB,
                   (const int*)w.colB, H, W, ntx, nty, per_plane, total);
        AMX_LAUNCH(locate_fold_kernel, dim3((unsigned)((mb / LOC_MB + 255) / 256)), dim3(256), 0, s, (const int*)w.L, w.cnt,
                   w.sr, w.sc, (const unsigned char*)w.rmap, mb, live, W);
    }
    const long nchunks = loc_nchunks(mb);
    AMX_LAUNCH(locate_count_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, (const int*)w.L,
               (const unsigned*)w.cnt, (const u64*)w.sr, (const u64*)w.sc, (const unsigned char*)w.rmap, mb, live,
               w.chunk_off, H, W, dist_edge);
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
    const long planes = (long)B * nch;
    const long live = planes * H * loc_wb(W), mb = loc_map_bytes(planes * H, W);
    const LocWork w = loc_views(const_cast<void*>(work), ne, mb, planes, H, W);
    AMX_LAUNCH(locate_emit_kernel, dim3((unsigned)loc_nchunks(mb)), dim3(256), 0, (hipStream_t)stream,
               (const int*)w.L, (const unsigned*)w.cnt, (const u64*)w.sr, (const u64*)w.sc,
               (const unsigned char*)w.rmap, mb, live, (const int*)w.chunk_off, coords, meta, cap, H, W, nch, dist_edge);
    AMX_CHECK_LAUNCH();
    return 0;
}
