// conv_ws.hip — wave-SPECIALISED plain 3x3 convolution for the thin layers (<= 32 input and <= 32 output channels at
// full / half resolution: U-Net's c2, c5b, c6 and their data gradients; atomai/nets/blocks.py:59-76, fcnn.py:100-138).
//
// Why a second kernel: on these layers conv_kernel.h is not limited by the matrix pipe but by everything around it —
// a wave spends 25 % of its life issuing MFMAs and the rest in index arithmetic, staging, epilogue and statistics
// (profiles/r02_conv_phases.md); four co-resident waves per SIMD in random phases leave the pipe idle whenever all four
// are outside their MFMA phase (1 - 0.75^4 = 0.68, the measured utilisation).  Here the two kinds of work run in
// DIFFERENT waves of one persistent 16-wave workgroup per CU (profiles/r03_micro_wave_specialised.md: a consumer /
// producer pair per SIMD keeps the pipe at 0.90 of the pure-MFMA ceiling in the micro-benchmark):
//
//   waves 0..7  (consumers): nothing but ds_read_b128 operand fetches + v_mfma_f32_16x16x4_f32 on a 16x16-pixel tile
//                            (2 image rows x NT cout tiles per wave — the inner loop of conv_kernel.h), then bias +
//                            LeakyReLU on the accumulators and a hand-over of the finished tile through LDS;
//   waves 8..15 (producers): global loads of tile k+2 into registers, BatchNorm affine + zero padding and the LDS image
//                            of tile k+1 (double-buffered), and the epilogue of tile k-1: per-strip batch statistics
//                            (sum, M2 about the strip mean — the same strips, rows and row order as conv_kernel.h, so
//                            bn.hip merges them unchanged) and 16-byte NHWC stores to one or two outputs.
//
// The weight image of the whole layer (<= 36 KB) is LDS-resident for the life of the workgroup; tiles are walked with a
// stride of gridDim.x so neighbouring tiles are in flight on neighbouring CUs (shared halo rows hit L2).  Two
// __syncthreads() per tile: A = "tile k computed, tile k+1 staged, hand-over buffer drained", B = "accumulators handed
// over".  LDS: weights + 2 input images + hand-over buffer = 156 KB for 32 -> 32 channels.
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

#define WS_IW 18                 // input image: (16 + 2) x (16 + 2) pixel slots
#define WS_SLOTS 324
#define WS_PLANE 336             // slots per 4-channel plane (== 0 mod 16: conflict-free b128 fragment reads)
#define WS_CONS 8                // consumer waves (2 image rows each); as many producer waves

template <int NCH, int NT>
__global__ __launch_bounds__(1024) void conv_ws_kernel(ConvFwdArgs a) {
    constexpr int COP = 16 * NT;
    constexpr int G = KG * NCH;                                   // 4-channel groups of the concatenated input
    constexpr int IN_FLOATS = G * WS_PLANE * 4;
    constexpr int W_FLOATS = NCH * 9 * KG * COP * 4;
    constexpr int HS = COP + 4;                                   // hand-over row stride: lane groups g land 16 banks apart
    constexpr int SH = NT == 2 ? 2 : 4;                           // rows of a statistics strip (plan_conv: th / 4)
    constexpr int CG = COP / 4;                                   // float4 groups per output pixel
    constexpr int XLD = (WS_SLOTS * G + 511) / 512;               // float4 loads per producer thread and tile
    AMX_DYN_SMEM(float, smem);
    float* s_w = smem;                                            // [NCH][9][KG][COP][4]
    float* s_in = smem + W_FLOATS;                                // [2][G][WS_PLANE][4]
    float* s_hand = s_in + 2 * IN_FLOATS;                         // [256 pixels][HS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = amx_wave_uniform(tid >> 6);
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    const int first = blockIdx.x, step = gridDim.x;
    const int my_tiles = (ntiles - first + step - 1) / step;      // >= 1: the launcher starts at most ntiles workgroups

    for (int i = tid; i < W_FLOATS / 4; i += 1024) amx_st4(s_w + 4 * i, amx_ld4(a.wpk + 4 * i));

    if (wave < WS_CONS) {
        // ------------------------------------------------------------------ consumers: operand reads + MFMA only
        const int p = lane & 15, g = lane >> 4;
        float bias_q[NT];
        #pragma unroll
        for (int q = 0; q < NT; ++q) bias_q[q] = a.bias ? a.bias[q * 16 + p] : 0.f;
        const float slope = a.slope;
        __syncthreads();                                          // weights + image of the first tile are in LDS
        for (int k = 0; k < my_tiles; ++k) {
            const float* in = s_in + (k & 1) * IN_FLOATS;
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
                        af[m] = amx_ld4(in + ((size_t)(chunk * KG + g) * WS_PLANE + (wave * 2 + m + 1 + dy) * WS_IW + (p + 1 + dx)) * 4);
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
            __syncthreads();                                      // A: the hand-over buffer is drained
            // C/D fragment: cout = lane & 15, pixel x = 4 * (lane >> 4) + reg
            #pragma unroll
            for (int m = 0; m < 2; ++m)
                #pragma unroll
                for (int q = 0; q < NT; ++q)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[m][q][r] + bias_q[q];
                        v = v > 0.f ? v : v * slope;
                        s_hand[(size_t)((wave * 2 + m) * TILE + 4 * g + r) * HS + q * 16 + p] = v;
                    }
            __syncthreads();                                      // B: handed over
        }
    } else {
        // ------------------------------------------------------------------ producers: loads, staging, epilogue
        const int ptid = tid - 64 * WS_CONS, pw = wave - WS_CONS;
        const int c8 = ptid % G;                                  // this thread's 4-channel group (512 % G == 0)
        const int slot0 = ptid / G;                               // its slots: slot0 + i * (512 / G)
        const int ch = c8 * 4;
        const float* src; int Cs, c;
        float4 r_sc = make_float4(1.f, 1.f, 1.f, 1.f), r_sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < a.C0s) { src = a.x0; Cs = a.C0s; c = ch; if (a.sc0) { r_sc = amx_ld4(a.sc0 + c); r_sh = amx_ld4(a.sh0 + c); } }
        else { src = a.x1; Cs = a.C1s; c = ch - a.C0s; if (a.sc1) { r_sc = amx_ld4(a.sc1 + c); r_sh = amx_ld4(a.sh1 + c); } }
        int rel[XLD], yx[XLD];                                    // pixel offset from the tile's halo origin; (iy << 8) | ix or -1
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int slot = slot0 + i * (512 / G);
            const int iy = slot / WS_IW, ix = slot - iy * WS_IW;
            rel[i] = iy * a.W + ix;
            yx[i] = slot < WS_SLOTS ? ((iy << 8) | ix) : -1;
        }
        float4 xr[XLD];
        unsigned xvalid = 0;
        auto tile_of = [&](int k, int& n, int& ty, int& tx) {
            int t = first + k * step;
            tx = t % a.tiles_x; t /= a.tiles_x;
            ty = t % a.tiles_y; n = t / a.tiles_y;
        };
        auto issue = [&](int k) {
            int n, ty, tx;
            tile_of(k, n, ty, tx);
            const int gy0 = ty * TILE - 1, gx0 = tx * TILE - 1;
            const long base = ((long)n * a.H + gy0) * a.W + gx0;
            xvalid = 0;
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                const int gy = gy0 + (yx[i] >> 8), gx = gx0 + (yx[i] & 255);
                const bool ok = yx[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) { xr[i] = amx_ld4(src + (size_t)(base + rel[i]) * Cs + c); xvalid |= 1u << i; }
            }
        };
        auto stage = [&](int buf) {
            float* dst = s_in + (size_t)buf * IN_FLOATS + (size_t)c8 * WS_PLANE * 4;
            #pragma unroll
            for (int i = 0; i < XLD; ++i) {
                if (yx[i] < 0) continue;
                float4 v = xr[i];
                if (xvalid & (1u << i)) {                         // the zero padding stays zero AFTER the affine
                    v.x = fmaf(v.x, r_sc.x, r_sh.x); v.y = fmaf(v.y, r_sc.y, r_sh.y);
                    v.z = fmaf(v.z, r_sc.z, r_sh.z); v.w = fmaf(v.w, r_sc.w, r_sh.w);
                }
                amx_st4(dst + (size_t)(slot0 + i * (512 / G)) * 4, v);
            }
        };
        auto epilogue = [&](int k) {
            if (pw >= TILE / SH) return;                          // one statistics strip (SH rows x 16 pixels) per wave
            int n, ty, tx;
            tile_of(k, n, ty, tx);
            float4 v[4];
            float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
            const int cg = lane % CG;
            const int co = cg * 4;
            #pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int ps = (it * 64 + lane) / CG;             // pixel of the strip: row ps / 16, column ps % 16
                const int row = pw * SH + ps / TILE, x = ps % TILE;
                v[it] = amx_ld4(s_hand + (size_t)(row * TILE + x) * HS + co);
                sm.x += v[it].x; sm.y += v[it].y; sm.z += v[it].z; sm.w += v[it].w;
                const size_t pix = ((size_t)n * a.H + ty * TILE + row) * a.W + tx * TILE + x;
                if (co < a.Y0s) amx_st4(a.y + pix * a.Y0s + co, v[it]);
                else amx_st4(a.y1 + pix * a.Y1s + (co - a.Y0s), v[it]);
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
        __syncthreads();
        for (int k = 0; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) {
                stage((k + 1) & 1);
                if (k + 2 < my_tiles) issue(k + 2);
            }
            if (k >= 1) epilogue(k - 1);
            __syncthreads();                                      // A
            __syncthreads();                                      // B
        }
        epilogue(my_tiles - 1);
    }
}

template <int NCH, int NT>
static int launch_conv_ws(const ConvFwdArgs& a, hipStream_t stream) {
    constexpr int COP = 16 * NT;
    const size_t lds = ((size_t)NCH * 9 * KG * COP * 4 + 2 * (size_t)KG * NCH * WS_PLANE * 4 + (size_t)256 * (COP + 4)) * sizeof(float);
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    int wgs = amx_num_cus();
    if (wgs > ntiles) wgs = ntiles;
#ifndef AMX_EMU
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_ws_kernel<NCH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
#endif
    AMX_LAUNCH((conv_ws_kernel<NCH, NT>), dim3(wgs), dim3(1024), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Which launches the wave-specialised kernel takes (everything else stays on conv_kernel.h): plain 3x3, 16 or 32
// concatenated input channels in whole 16-channel chunks per source, 16 or 32 output channels stored without padding,
// image sides in multiples of 16 and enough tiles for one persistent workgroup per CU; no fused head / block sum /
// residual addend / post-affine activation.  `strip` is the statistics strip height the layer's plan reports.
bool amx_conv_ws_supported(const ConvFwdArgs& a, int taps, int dil, int strip, float in_slope0, float in_slope1) {
    static int enabled = -1;
    if (const char* e = getenv("AMX_CONV_WS")) enabled = atoi(e) != 0; else if (enabled < 0) enabled = 1;
    if (!enabled) return false;
    if (taps != 9 || dil != 1 || a.hout || a.nds || a.addend || in_slope0 != 1.f || in_slope1 != 1.f) return false;
    const int cin = a.C0s + a.C1s;
    if ((cin != 16 && cin != 32) || (a.C0s & 15) || (a.C1s & 15)) return false;
    if ((a.cout != 16 && a.cout != 32) || a.Y0s + a.Y1s != a.cout) return false;
    if ((a.H & 15) || (a.W & 15)) return false;
    if (strip != (a.cout == 32 ? 2 : 4)) return false;
    if ((long)(a.H / 16) * (a.W / 16) * a.N < 2L * amx_num_cus()) return false;
    return true;
}

static long ws_launches = 0;
extern "C" long amx_conv2d_ws_launches(void) { return ws_launches; }

int amx_conv_launch_ws(ConvFwdArgs& a, hipStream_t s) {
    ++ws_launches;
    a.tiles_x = a.W / TILE; a.tiles_y = a.H / TILE;
    const int nch = (a.C0s + a.C1s) / 16;
    if (nch == 1) return a.cout == 16 ? launch_conv_ws<1, 1>(a, s) : launch_conv_ws<1, 2>(a, s);
    return a.cout == 16 ? launch_conv_ws<2, 1>(a, s) : launch_conv_ws<2, 2>(a, s);
}
