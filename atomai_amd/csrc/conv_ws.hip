// conv_ws.hip — wave-SPECIALISED plain 3x3 convolution for the thin layers (<= 32 input and <= 32 output channels at
// full / half resolution: U-Net's c2, c5b, c6 and their data gradients; atomai/nets/blocks.py:59-76, fcnn.py:100-138).
//
// Why a second kernel: on these layers conv_kernel.h is not limited by the matrix pipe but by everything around it —
// a wave spends 25 % of its life issuing MFMAs and the rest in index arithmetic, staging, epilogue and statistics
// (profiles/r02_conv_phases.md); four co-resident waves per SIMD in random phases leave the pipe idle whenever all four
// are outside their MFMA phase (1 - 0.75^4 = 0.68, the measured utilisation).  Here the two kinds of work run in
// DIFFERENT waves of one persistent 16-wave workgroup per CU (profiles/r03_wave_specialised.md: a consumer /
// producer pair per SIMD keeps the pipe at 0.90 of the pure-MFMA ceiling in the micro-benchmark):
//
//   waves 0..7  (consumers): nothing but ds_read_b128 operand fetches + v_mfma_f32_16x16x4_f32 on a 16x16-pixel tile
//                            (2 image rows x NT cout tiles per wave — the inner loop of conv_kernel.h), then ONE b128
//                            store per accumulator tile into a cout-major hand-over buffer (the four registers of a
//                            C/D fragment are four consecutive pixels of one cout);
//   waves 8..15 (producers, s_setprio 3): global loads of tile k+2 into registers, BatchNorm affine + zero padding and
//                            the LDS image of tile k+1 (double-buffered), and the epilogue of tile k-1: bias +
//                            LeakyReLU, per-strip batch statistics (sum, M2 about the strip mean — the same strips,
//                            rows and row order as conv_kernel.h, so bn.hip merges them unchanged) and 16-byte NHWC
//                            stores to one or two outputs.
//
// The weight image of the whole layer (<= 36 KB) is LDS-resident for the life of the workgroup; tiles are walked with a
// stride of gridDim.x so neighbouring tiles are in flight on neighbouring CUs (shared halo rows hit L2).  Two
// __syncthreads() per tile: A = "tile k computed, tile k+1 staged, hand-over buffer drained", B = "accumulators handed
// over".  LDS: weights + 2 input images + hand-over buffer = 153 KB for 32 -> 32 channels.
//
// What the phase clocks showed while it was built (tools/gpu_ws_phases.py, profiles/r03_wave_specialised.md):
//   * without raised priority the producers only advance when the consumers stop at a barrier (the issue arbiter serves
//     the oldest ready wave and a consumer always has an MFMA waiting for the pipe): staging took "sweep + 1.8 k clocks";
//   * a predicated load, or separate interior / border code paths, makes the compiler copy loaded registers at the
//     merge and wait for them where they were issued: every load here is unconditional (clamped address, zeroed later);
//   * the 64-bit address arithmetic of 10 accesses per tile is worth precomputing: the producers share the VALU port
//     with the MFMAs and get about one slot in 10-16 clocks.
//
// Same arguments, results and partial-statistics layout as amx_conv2d_fwd / amx_conv2d_dgrad; conv_fwd.hip routes a
// launch here when amx_conv_ws_supported() says so (AMX_CONV_WS=0 switches it off for A/B measurements).
#include "conv_kernel.h"

static __device__ __forceinline__ int amx_wave_uniform(int v) {
#ifdef AMX_EMU
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// AMX_CONV_PROFILE (dev builds, tools/gpu_ws_phases.py): every wave accumulates the shader clocks of its phases over all its
// tiles and writes [workgroup][wave][8] totals: 0 MFMA sweep, 1 wait at barrier A, 2 hand-over, 3 wait at barrier B,
// 4 staging (incl. the wait for the tile's loads), 5 issuing loads, 6 epilogue, 7 lifetime.
#ifdef AMX_CONV_PROFILE
#define WS_TICK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt[i] += n_ - pl; pl = n_; } while (0)
#else
#define WS_TICK(i) do { } while (0)
#endif

#ifndef AMX_WS_PRODUCER_PRIO
#define AMX_WS_PRODUCER_PRIO 3   // s_setprio of the producer waves (0 = leave the default; experiment switch)
#endif
#ifndef AMX_WS_STAGE_MAP
#define AMX_WS_STAGE_MAP 0
#endif
#define WS_IW 18                 // input image: (16 + 2) x (16 + 2) pixel slots
#define WS_SLOTS 324
#define WS_CONS 8                // consumer waves (2 image rows each); as many producer waves
#define WS_PS 260                // hand-over buffer [cout][256 pixels + 4]: the 16 couts of a b128 store land 4 banks apart

// slots (16 B) per 4-channel plane of the input image: the staging stores of 16 consecutive lanes (channel group fastest,
// then slot) must cover all 64 banks -> plane stride == 8 (two chunks: 8 groups x 2 slots) or 16 (one chunk: 4 x 4) mod 64
// floats; the consumers' fragment reads are contiguous within a plane and do not care
template <int NCH> struct WsPlane { static constexpr int value = NCH == 2 ? 338 : 340; };

// BWD (round 4): the data-gradient launch of a layer with BatchNorm — x0 is dy, the gradient w.r.t. the layer's output, and
// the producers form  dpre = lrelu'(a) * (k1 * dy + k2 * a + k3)  from (dy, a) while staging (ConvFwdArgs::bw_*; the
// arithmetic of amx_bn_bwd_apply, so the input image — and hence the result — is bit-identical to the two-pass form).
// The separate amx_bn_bwd_apply pass (read dy, read a, write dpre: 0.16-0.32 ms per thin layer, on the critical path of
// the backward pass) disappears; the producers' second load stream costs registers only they need.
// BSUM (round 6, data-gradient launches with ONE output whose position is the output of a conv -> LeakyReLU -> BatchNorm
// layer): the accumulators a CONSUMER wave hands over are that layer's dy (a data gradient has no bias and no activation), so
// the consumer also keeps per-lane running sums of dy and dy * a over all tiles of the workgroup — the saved activation a of
// the tile's pixels is requested at the start of the sweep, in the C/D fragment layout (lane = cout, four consecutive
// pixels), and lands behind the MFMAs; the adds / fmas fill issue slots in the shadow of the matrix pipe — and one row of
// (sum dy, sum dy * a) per consumer wave leaves at the end (a.bs_part).  amx_bn_bwd_reduce — a pass over both tensors on
// the critical chain of the backward pass — is not launched for that layer.  The producers could not carry it: four more
// float4 in flight next to their two register images of tile k + 2 spilled ~30 registers.  Fixed tile -> workgroup -> wave
// -> lane assignment: deterministic.
template <int NCH, int NT, bool BWD, bool BSUM = false>
__global__ __launch_bounds__(1024) void conv_ws_kernel(ConvFwdArgs a) {
    constexpr int COP = 16 * NT;
    constexpr int G = KG * NCH;                                   // 4-channel groups of the concatenated input
    constexpr int PLANE = WsPlane<NCH>::value;
    constexpr int IN_FLOATS = G * PLANE * 4;
    constexpr int W_FLOATS = NCH * 9 * KG * COP * 4;
    constexpr int SH = NT == 2 ? 2 : 4;                           // rows of a statistics strip (plan_conv: th / 4)
    constexpr int CG = COP / 4;                                   // float4 groups per output pixel
    constexpr int XLD = (WS_SLOTS * G + 511) / 512;               // float4 loads per producer thread and tile
    AMX_DYN_SMEM(float, smem);
    float* s_w = smem;                                            // [NCH][9][KG][COP][4]
    float* s_in = smem + W_FLOATS;                                // [2][G][PLANE][4]
    float* s_hand = s_in + 2 * IN_FLOATS;                         // [COP][WS_PS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = amx_wave_uniform(tid >> 6);
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    const int first = blockIdx.x, step = gridDim.x;
    const int my_tiles = (ntiles - first + step - 1) / step;      // >= 1: the launcher starts at most ntiles workgroups

#ifdef AMX_CONV_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pl = __builtin_amdgcn_s_memtime();
    const unsigned long long p0 = pl;
#endif
    for (int i = tid; i < W_FLOATS / 4; i += 1024) amx_st4(s_w + 4 * i, amx_ld4(a.wpk + 4 * i));

    if (wave < WS_CONS) {
        // ------------------------------------------------------------------ consumers: operand reads + MFMA only
        const int p = lane & 15, g = lane >> 4;
        __syncthreads();                                          // weights + image of the first tile are in LDS
        WS_TICK(3);
        float bs1[NT], bs2[NT];                                   // BSUM: running sum dy, sum dy * a of cout q * 16 + p
        #pragma unroll
        for (int q = 0; q < NT; ++q) { bs1[q] = 0.f; bs2[q] = 0.f; }
        for (int k = 0; k < my_tiles; ++k) {
            const float* in = s_in + (k & 1) * IN_FLOATS;
            float av[BSUM ? 2 : 1][NT][4];
            if (BSUM) {
                // saved activation of this wave's 2 rows x 16 pixels x COP couts, element (row m, pixel 4g + r, cout q*16 + p)
                int t = first + k * step;
                const int tx = t % a.tiles_x; t /= a.tiles_x;
                const int ty = t % a.tiles_y; const int n = t / a.tiles_y;
                const float* ab = a.bs_a + ((size_t)(n * a.H + ty * TILE + wave * 2) * a.W + tx * TILE + 4 * g) * COP + p;
                #pragma unroll
                for (int m = 0; m < 2; ++m)
                    #pragma unroll
                    for (int q = 0; q < NT; ++q)
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) av[m][q][r] = ab[((size_t)m * a.W + r) * COP + q * 16];
            }
            f32x4 acc[2][NT];
            #pragma unroll
            for (int m = 0; m < 2; ++m)
                #pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int chunk = 0; chunk < NCH; ++chunk) {
                #pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    float4 af[2], bf[NT];
                    #pragma unroll
                    for (int m = 0; m < 2; ++m)
                        af[m] = amx_ld4(in + ((size_t)(chunk * KG + g) * PLANE + (wave * 2 + m + 1 + dy) * WS_IW + (p + 1 + dx)) * 4);
                    #pragma unroll
                    for (int q = 0; q < NT; ++q)
                        bf[q] = amx_ld4(s_w + ((size_t)((chunk * 9 + tap) * KG + g) * COP + q * 16 + p) * 4);
                    #define WS_MFMA(C)                                                                    \
                        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                     \
                            _Pragma("unroll") for (int q = 0; q < NT; ++q)                                \
                                acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].C, bf[q].C, acc[m][q], 0, 0, 0);
                    WS_MFMA(x) WS_MFMA(y) WS_MFMA(z) WS_MFMA(w)
                    #undef WS_MFMA
                }
            }
            if (BSUM) {
                #pragma unroll
                for (int m = 0; m < 2; ++m)
                    #pragma unroll
                    for (int q = 0; q < NT; ++q)
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) { bs1[q] += acc[m][q][r]; bs2[q] = fmaf(acc[m][q][r], av[m][q][r], bs2[q]); }
            }
            WS_TICK(0);
            __syncthreads();                                      // A: the hand-over buffer is drained
            WS_TICK(1);
            // C/D fragment: cout = lane & 15, pixels x = 4 * (lane >> 4) + reg: the four registers are four CONSECUTIVE
            // pixels of one cout -> one b128 store per accumulator tile into the cout-major hand-over buffer
            #pragma unroll
            for (int m = 0; m < 2; ++m)
                #pragma unroll
                for (int q = 0; q < NT; ++q)
                    amx_st4(s_hand + (size_t)(q * 16 + p) * WS_PS + (wave * 2 + m) * TILE + 4 * g,
                            make_float4(acc[m][q][0], acc[m][q][1], acc[m][q][2], acc[m][q][3]));
            WS_TICK(2);
            __syncthreads();                                      // B: handed over
            WS_TICK(3);
        }
        if (BSUM) {
            // the four lane groups g hold partial sums of the same cout p: butterfly, then one row per consumer wave
            #pragma unroll
            for (int q = 0; q < NT; ++q) {
                bs1[q] += __shfl_xor(bs1[q], 16); bs1[q] += __shfl_xor(bs1[q], 32);
                bs2[q] += __shfl_xor(bs2[q], 16); bs2[q] += __shfl_xor(bs2[q], 32);
                if (g == 0) {
                    const size_t row = (size_t)blockIdx.x * WS_CONS + wave;
                    a.bs_part[(row * 2) * COP + q * 16 + p] = bs1[q];
                    a.bs_part[(row * 2 + 1) * COP + q * 16 + p] = bs2[q];
                }
            }
        }
    } else {
        // ------------------------------------------------------------------ producers: loads, staging, epilogue
        // Everything a producer does shares the SIMD's VALU issue port with the consumers' MFMAs (about one slot per
        // 10-16 clocks next to a saturated matrix pipe, profiles/r03_wave_specialised.md), so the per-tile
        // instruction count is what matters here: per-thread byte offsets are computed once, a tile costs one 64-bit
        // multiply-add per base pointer plus one 64-bit add per access, and tiles that do not touch the image border
        // skip every bounds predicate.
#if !defined(AMX_EMU) && AMX_WS_PRODUCER_PRIO
        // The issue arbiter serves the oldest ready wave first, and a consumer's next MFMA waiting for the busy matrix pipe
        // sits in front of a producer's VALU / LDS / memory instructions; raised priority lets the few producer
        // instructions through while the pipe works off its queue.
        __builtin_amdgcn_s_setprio(AMX_WS_PRODUCER_PRIO);
#endif
        const int ptid = tid - 64 * WS_CONS, pw = wave - WS_CONS;
        // AMX_WS_STAGE_MAP (experiment switch, round 6): 0 = channel group fastest — 8 consecutive lanes stage 8 / G slots of
        // G planes, a 2-way conflict of the ds_write_b128 (15 instead of 8 LDS cycles per wave instruction,
        // profiles/r06_lds_conflicts.md `ws_stage_write`); 1 = 8 consecutive lanes take 8 consecutive slots of one plane
#if AMX_WS_STAGE_MAP
        const int c8 = (ptid >> 3) % G;
        const int slot0 = (ptid & 7) + 8 * (ptid / (8 * G));
#else
        const int c8 = ptid % G;                                  // this thread's 4-channel group (512 % G == 0)
        const int slot0 = ptid / G;                               // its slots: slot0 + i * (512 / G)
#endif
        const int ch = c8 * 4;
        const char* src; unsigned cs_bytes;                       // this thread's source (bytes) at its channel group
        float4 r_sc = make_float4(1.f, 1.f, 1.f, 1.f), r_sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < a.C0s) {
            src = (const char*)(a.x0 + ch); cs_bytes = (unsigned)a.C0s * 4u;
            if (a.sc0) { r_sc = amx_ld4(a.sc0 + ch); r_sh = amx_ld4(a.sh0 + ch); }
        } else {
            src = (const char*)(a.x1 + (ch - a.C0s)); cs_bytes = (unsigned)a.C1s * 4u;
            if (a.sc1) { r_sc = amx_ld4(a.sc1 + (ch - a.C0s)); r_sh = amx_ld4(a.sh1 + (ch - a.C0s)); }
        }
        // (the three per-channel constants of the on-load backward are re-read from L1 when a tile is staged: holding them
        //  next to two 6-slot register images pushed the 32-channel class over the 128 registers a 16-wave workgroup has)
        const long aux_delta = BWD ? (const char*)a.bw_aux - (const char*)a.x0 : 0;      // bytes from dy to the saved activation
        const float bslope = a.bw_slope;
        int yx[XLD];                                              // (iy << 8) | ix of the thread's slots (the last may be past the image: -1)
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int slot = slot0 + i * (512 / G);
            const int iy = slot / WS_IW, ix = slot - iy * WS_IW;
            yx[i] = slot < WS_SLOTS ? ((iy << 8) | ix) : -1;
        }
        float4 xr[XLD];
        float4 ar[BWD ? XLD : 1];
        unsigned xvalid = 0;
        auto tile_of = [&](int k, int& n, int& ty, int& tx) {
            int t = first + k * step;
            tx = t % a.tiles_x; t /= a.tiles_x;
            ty = t % a.tiles_y; n = t / a.tiles_y;
        };
        // Every load is UNCONDITIONAL: a predicated load (or two code paths for interior / border tiles) makes the
        // compiler merge old and new register values, and the copy it inserts waits for the load right where it was
        // issued — the whole prefetch distance is lost.  Halo slots outside the image read a clamped (valid) address
        // instead and are zeroed when the tile is staged.
        auto issue = [&](int k) {
            int n, ty, tx;
            tile_of(k, n, ty, tx);
            const int ylo = ty == 0 ? 1 : 0, yhi = ty + 1 == a.tiles_y ? TILE : TILE + 1;     // valid slot rows / columns
            const int xlo = tx == 0 ? 1 : 0, xhi = tx + 1 == a.tiles_x ? TILE : TILE + 1;
            const long origin = ((long)n * a.H + ty * TILE - 1) * a.W + tx * TILE - 1;        // halo origin (may be < 0)
            const char* base = src + origin * (long)cs_bytes;
            xvalid = 0;
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int iy = yx[i] >> 8, ix = yx[i] & 255;      // (-1 -> row -1, column 255: clamped like any other)
                const int cy = min(max(iy, ylo), yhi), cx = min(max(ix, xlo), xhi);
                if (cy == iy && cx == ix) xvalid |= 1u << i;
                const char* pa = base + (unsigned)(cy * a.W + cx) * cs_bytes;
                xr[i] = *reinterpret_cast<const float4*>(pa);
                if (BWD) ar[i] = *reinterpret_cast<const float4*>(pa + aux_delta);
            }
        };
        auto stage = [&](int buf) {
            float* dst = s_in + (size_t)buf * IN_FLOATS + (size_t)c8 * PLANE * 4 + (size_t)slot0 * 4;
            float4 c1 = make_float4(1.f, 1.f, 1.f, 1.f), c2 = make_float4(0.f, 0.f, 0.f, 0.f), c3 = c2;
            if (BWD && a.bw_k1) { c1 = amx_ld4(a.bw_k1 + ch); c2 = amx_ld4(a.bw_k2 + ch); c3 = amx_ld4(a.bw_k3 + ch); }
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                if (i + 1 == XLD && yx[i] < 0) continue;          // (only the last load of a thread can fall off the image)
                const float4 v = xr[i];
                float4 w;
                if (BWD) {
                    const float4 t = ar[i];
                    w.x = (t.x > 0.f ? 1.f : bslope) * fmaf(c1.x, v.x, fmaf(c2.x, t.x, c3.x));
                    w.y = (t.y > 0.f ? 1.f : bslope) * fmaf(c1.y, v.y, fmaf(c2.y, t.y, c3.y));
                    w.z = (t.z > 0.f ? 1.f : bslope) * fmaf(c1.z, v.z, fmaf(c2.z, t.z, c3.z));
                    w.w = (t.w > 0.f ? 1.f : bslope) * fmaf(c1.w, v.w, fmaf(c2.w, t.w, c3.w));
                } else {
                    w = make_float4(fmaf(v.x, r_sc.x, r_sh.x), fmaf(v.y, r_sc.y, r_sh.y),
                                    fmaf(v.z, r_sc.z, r_sh.z), fmaf(v.w, r_sc.w, r_sh.w));
                }
                if (!(xvalid & (1u << i))) w = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding (AFTER the affine)
                amx_st4(dst + (size_t)i * (512 / G) * 4, w);
            }
        };
        // epilogue: wave pw owns statistics strip pw (SH rows x 16 pixels) of the tile; lane -> (pixel, 4 couts)
        const int cg = lane % CG, co = cg * 4;
        const bool epi_wave = pw < TILE / SH;
        char* outp; unsigned cd_bytes;                            // this lane's output (bytes) at its cout group
        if (co < a.Y0s) { outp = (char*)(a.y + co); cd_bytes = (unsigned)a.Y0s * 4u; }
        else { outp = (char*)(a.y1 + (co - a.Y0s)); cd_bytes = (unsigned)a.Y1s * 4u; }
        unsigned orel[4]; int hoff[4];
        #pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ps = (it * 64 + lane) / CG;                 // pixel of the strip: row ps / 16, column ps % 16
            const int row = (epi_wave ? pw : 0) * SH + ps / TILE, x = ps % TILE;
            orel[it] = (unsigned)(row * a.W + x) * cd_bytes;
            hoff[it] = co * WS_PS + row * TILE + x;
        }
        const float4 b4 = a.bias ? amx_ld4(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float slope = a.slope;
        auto epilogue = [&](int k) {
            if (!epi_wave) return;
            int n, ty, tx;
            tile_of(k, n, ty, tx);
            char* obase = outp + (((long)n * a.H + ty * TILE) * a.W + tx * TILE) * (long)cd_bytes;
            float4 v[4];
            float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
            #pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float* h = s_hand + hoff[it];
                float4 t = make_float4(h[0] + b4.x, h[WS_PS] + b4.y, h[2 * WS_PS] + b4.z, h[3 * WS_PS] + b4.w);
                t.x = t.x > 0.f ? t.x : t.x * slope; t.y = t.y > 0.f ? t.y : t.y * slope;
                t.z = t.z > 0.f ? t.z : t.z * slope; t.w = t.w > 0.f ? t.w : t.w * slope;
                v[it] = t;
                sm.x += t.x; sm.y += t.y; sm.z += t.z; sm.w += t.w;
                *reinterpret_cast<float4*>(obase + orel[it]) = t;
            }
            if (a.stats) {
                #pragma unroll
                for (int o = CG; o < 64; o <<= 1) {
                    sm.x += __shfl_xor(sm.x, o); sm.y += __shfl_xor(sm.y, o); sm.z += __shfl_xor(sm.z, o); sm.w += __shfl_xor(sm.w, o);
                }
                const float inv = 1.0f / (float)(SH * TILE);
                const float4 mu = make_float4(sm.x * inv, sm.y * inv, sm.z * inv, sm.w * inv);
                float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
                #pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float dx = v[it].x - mu.x, dy = v[it].y - mu.y, dz = v[it].z - mu.z, dw = v[it].w - mu.w;
                    s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
                }
                #pragma unroll
                for (int o = CG; o < 64; o <<= 1) {
                    s2.x += __shfl_xor(s2.x, o); s2.y += __shfl_xor(s2.y, o); s2.z += __shfl_xor(s2.z, o); s2.w += __shfl_xor(s2.w, o);
                }
                if (lane < CG) {                                  // rows [n][strip][tx] as conv_kernel.h writes them
                    const size_t row = ((size_t)n * (a.H / SH) + ty * (TILE / SH) + pw) * a.tiles_x + tx;
                    amx_st4(a.stats + (row * 2) * COP + co, sm);
                    amx_st4(a.stats + (row * 2 + 1) * COP + co, s2);
                }
            }
        };
        issue(0);
        stage(0);
        if (my_tiles > 1) issue(1);
        WS_TICK(4);
        __syncthreads();
        WS_TICK(3);
        for (int k = 0; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) {
                stage((k + 1) & 1);
                WS_TICK(4);
                if (k + 2 < my_tiles) issue(k + 2);
                WS_TICK(5);
            }
            if (k >= 1) epilogue(k - 1);
            WS_TICK(6);
            __syncthreads();                                      // A
            WS_TICK(1);
            __syncthreads();                                      // B
            WS_TICK(3);
        }
        epilogue(my_tiles - 1);
        WS_TICK(6);
    }
#ifdef AMX_CONV_PROFILE
    if (a.prof && lane == 0) {
        pt[7] = __builtin_amdgcn_s_memtime() - p0;
        for (int i = 0; i < 8; ++i) a.prof[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + i] = pt[i];
    }
#endif
}

template <int NCH, int NT, bool BWD, bool BSUM = false>
static int launch_conv_ws(const ConvFwdArgs& a, hipStream_t stream) {
    constexpr int COP = 16 * NT;
    const size_t lds = ((size_t)NCH * 9 * KG * COP * 4 + 2 * (size_t)KG * NCH * WsPlane<NCH>::value * 4 + (size_t)COP * WS_PS) * sizeof(float);
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    int wgs = amx_num_cus();
    if (wgs > ntiles) wgs = ntiles;
    AMX_ALLOW_160K_LDS(conv_ws_kernel<NCH, NT, BWD, BSUM>);
    AMX_LAUNCH((conv_ws_kernel<NCH, NT, BWD, BSUM>), dim3(wgs), dim3(1024), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Which launches the wave-specialised kernel takes (everything else stays on conv_kernel.h): plain 3x3, 16 or 32
// concatenated input channels in whole 16-channel chunks per source, 16 or 32 output channels stored without padding,
// image sides in multiples of 16 and enough tiles for one persistent workgroup per CU; no fused head / block sum /
// residual addend / post-affine activation.  `strip` is the statistics strip height the layer's plan reports.
bool amx_conv_ws_supported(const ConvFwdArgs& a, int taps, int dil, int strip, float in_slope0, float in_slope1) {
    // AMX_CONV_WS: 0 = off, 1 (default) = forward launches + the data-gradient classes of AMX_CONV_WS_DGRAD, 2 = forward
    // only (a bias is present), 3 = data gradients only (every class).  AMX_CONV_WS_DGRAD is a mask of data-gradient
    // classes: 1 = two-output launches (the data gradient of a layer that read a concatenation — U-Net c6.0: 16 -> 16 + 16,
    // the FIRST data gradient of the backward pass), 2 = 32 -> 32 channels (c5.3, c2.3), 4 = the rest (32 -> 16: c2.0, the
    // last one, which finds the side stream's backlog of weight gradients in its way); default 7 since round 4: with the
    // layer's BatchNorm backward formed by this kernel's loader (BWD) every class saves its amx_bn_bwd_apply pass, and the
    // 32 -> 16 launch, a 0.09 ms loss in round 3, becomes a gain (17.95 -> 17.80 ms, profiles/r04_logs/r04_bwd_fuse_ab.log).
    // Stand-alone the kernel is 12-27 % faster than conv_kernel.h on every thin shape, forward and data gradient alike
    // (profiles/r03_wave_specialised.md).  Inside the training step a persistent 16-wave workgroup owns its CU's LDS and
    // registers, so the weight-gradient kernels of the side stream cannot run next to a wave-specialised data gradient
    // and the lost overlap competes with what the faster kernel wins; measured per class in-process
    // (profiles/r03_dgrad_first_ab.log): forward only 18.17 ms, + c6.0 18.03, + the two 32 -> 32 launches 17.99, while the
    // 32 -> 16 launch costs 0.09 ms (all classes: 18.1-18.5, the round's earlier "no gain" result).
    const int mode = amx_knobs().conv_ws, dmask = amx_knobs().conv_ws_dgrad;
    if (mode <= 0) return false;
    if ((mode == 2 && !a.bias) || (mode == 3 && a.bias)) return false;
    if (mode == 1 && !a.bias) {
        const int cls = a.Y1s > 0 ? 1 : ((a.C0s + a.C1s == 32 && a.cout == 32) ? 2 : 4);
        if (!(dmask & cls)) return false;
    }
    if (a.bw_aux && (a.bias || a.C1s || a.sc0 || a.stats)) return false;      // the on-load backward: one plain dy source
    if (taps != 9 || dil != 1 || a.hout || a.nds || a.addend || in_slope0 != 1.f || in_slope1 != 1.f) return false;
    const int cin = a.C0s + a.C1s;
    if ((cin != 16 && cin != 32) || (a.C0s & 15) || (a.C1s & 15)) return false;
    if ((a.cout != 16 && a.cout != 32) || a.Y0s + a.Y1s != a.cout) return false;
    if ((a.H & 15) || (a.W & 15)) return false;
    if (strip != (a.cout == 32 ? 2 : 4)) return false;
    if ((long)(a.H / 16) * (a.W / 16) * a.N < 2L * amx_num_cus()) return false;
    return true;
}

// rows of the BatchNorm-backward partial sums a BSUM launch writes: one per workgroup and consumer wave
int amx_conv_ws_bsum_rows(int N, int H, int W, int cout) {
    const long ntiles = (long)(H / TILE) * (W / TILE) * N;
    const int wgs = (int)(ntiles < amx_num_cus() ? ntiles : amx_num_cus());
    return wgs * WS_CONS;
}

static std::atomic<long> ws_launches{0};
extern "C" long amx_conv2d_ws_launches(void) { return ws_launches.load(std::memory_order_relaxed); }

int amx_conv_launch_ws(ConvFwdArgs& a, hipStream_t s) {
    ws_launches.fetch_add(1, std::memory_order_relaxed);
    a.tiles_x = a.W / TILE; a.tiles_y = a.H / TILE;
    const int nch = (a.C0s + a.C1s) / 16;
    if (a.bw_aux && a.bs_a) {                  // + BatchNorm-backward sums of the output position's layer (one output)
        if (nch == 1) return a.cout == 16 ? launch_conv_ws<1, 1, true, true>(a, s) : launch_conv_ws<1, 2, true, true>(a, s);
        return a.cout == 16 ? launch_conv_ws<2, 1, true, true>(a, s) : launch_conv_ws<2, 2, true, true>(a, s);
    }
    if (a.bw_aux) {
        if (nch == 1) return a.cout == 16 ? launch_conv_ws<1, 1, true>(a, s) : launch_conv_ws<1, 2, true>(a, s);
        return a.cout == 16 ? launch_conv_ws<2, 1, true>(a, s) : launch_conv_ws<2, 2, true>(a, s);
    }
    if (nch == 1) return a.cout == 16 ? launch_conv_ws<1, 1, false>(a, s) : launch_conv_ws<1, 2, false>(a, s);
    return a.cout == 16 ? launch_conv_ws<2, 1, false>(a, s) : launch_conv_ws<2, 2, false>(a, s);
}
