// upconv.hip — UpsampleBlock forward in ONE pass (round 5).
//
//   UpsampleBlock.forward: F.interpolate(x, scale_factor=2, mode) -> Conv2d 1x1 (+bias)          atomai/nets/blocks.py:122-132
// The 1x1 convolution commutes with the interpolation (its weights sum to one), so it is evaluated at LOW resolution
// (DESIGN.md §2).  Rounds 1-4 ran it as a conv_kernel.h launch that wrote the low-resolution result v and a second kernel
// (spatial.hip: upsample_fwd_kernel) that read v back and wrote the x2 tensor.  Here one workgroup owns a 6 x 14 low-res
// tile: it forms v on its 8 x 16 halo'd pixel set with fp32 MFMAs straight from global memory (transposed GEMM
// D'[cout][pixel] = W x^T, the weights of the layer in LDS), leaves v in LDS and writes the 12 x 28 high-resolution outputs
// from there — v never exists in HBM, one launch instead of two.
//   * same k order as conv_kernel.h's 1x1 class (channel groups ascending, 4 k per MFMA)  -> the same v, bit for bit;
//   * the interpolation is upsample_fwd_kernel's expression (rows first, then columns; indices clamped = halo pixels are
//     evaluated at clamped coordinates)                                                      -> the same output, bit for bit.
// The BatchNorm affine of the producer (scale, shift per input channel) is applied on load, as everywhere.
#include "amx_device.h"

#define UC_TX 14                 // interior low-res columns per workgroup (+2 halo = 16 = one MFMA pixel tile)
#define UC_COLS (UC_TX + 2)
// rows: RW halo rows per wave (2 or 4) -> 4 RW halo rows, 4 RW - 2 interior rows per workgroup (6 x 14 or 14 x 14 pixels)

#ifndef UPCONV_RW4
#define UPCONV_RW4 1            // compile-time A/B switch: 0 = 6 x 14 tiles for every class
#endif

struct UpConvArgs {
    const float* x; const float* sc; const float* sh; const float* w; const float* bias; float* y;
    int N, h, w_, Cin, Cs_in, Cout, Cs_out, mode, tiles_x, tiles_y;
};

template <int NTC, int RW>       // cout tiles of 16 (1, 2 or 4); halo rows per wave (2 or 4)
__global__ __launch_bounds__(256) void upconv1x1_fwd_kernel(UpConvArgs a) {
    constexpr int COP = 16 * NTC;
    constexpr int UC_ROWS = 4 * RW, UC_TY = UC_ROWS - 2;
    constexpr int VP = COP + 4;                                   // floats per pixel slot of the v image
    AMX_DYN_SMEM(float, smem);
    const int KS = amx_round_up(a.Cs_in, 16), WP = KS + 4;        // floats per cout row of the weight image
    float* s_w = smem;                                            // [COP][WP]
    float* s_v = smem + (size_t)COP * WP;                         // [UC_ROWS * 16][VP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int KS4 = KS >> 2;
    if ((a.Cin & 3) == 0) {                                       // rows of whole float4s (every reference net but dilnet's 50)
        for (int i = tid; i < COP * KS4; i += 256) {
            const int co = i / KS4, k = 4 * (i - co * KS4);
            const float4 v = (co < a.Cout && k < a.Cin) ? amx_ld4(a.w + (size_t)co * a.Cin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            amx_st4(s_w + (size_t)co * WP + k, v);
        }
    } else {
        for (int i = tid; i < COP * KS; i += 256) {
            const int co = i / KS, k = i - co * KS;
            s_w[(size_t)co * WP + k] = (co < a.Cout && k < a.Cin) ? a.w[(size_t)co * a.Cin + k] : 0.f;
        }
    }
    const int ntiles = a.tiles_x * a.tiles_y * a.N;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int bx = t % a.tiles_x; t /= a.tiles_x;
    const int by = t % a.tiles_y; const int n = t / a.tiles_y;
    const int ty0 = by * UC_TY, tx0 = bx * UC_TX;                 // first interior low-res pixel of the tile

    // this lane's two pixels (halo rows 2 wave, 2 wave + 1; halo column p), at clamped image coordinates
    const int nkg = a.Cs_in >> 2, nchunk = (nkg + 3) >> 2;
    const float* px[RW];
    #pragma unroll
    for (int m = 0; m < RW; ++m) {
        int yy = ty0 - 1 + RW * wave + m, xx = tx0 - 1 + p;
        yy = yy < 0 ? 0 : (yy > a.h - 1 ? a.h - 1 : yy);
        xx = xx < 0 ? 0 : (xx > a.w_ - 1 ? a.w_ - 1 : xx);
        px[m] = a.x + (((size_t)n * a.h + yy) * a.w_ + xx) * a.Cs_in;
    }
    auto load = [&](int c, float4* b) {
        const int kg = 4 * c + g;
        const bool ok = kg < nkg;
        const int off = ok ? 4 * kg : 0;                          // (unconditional loads at a clamped offset, zeroed below)
        float4 s1 = make_float4(1.f, 1.f, 1.f, 1.f), s0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.sc) { s1 = amx_ld4(a.sc + off); s0 = amx_ld4(a.sh + off); }
        #pragma unroll
        for (int m = 0; m < RW; ++m) {
            float4 v = amx_ld4(px[m] + off);
            if (a.sc) { v.x = fmaf(v.x, s1.x, s0.x); v.y = fmaf(v.y, s1.y, s0.y); v.z = fmaf(v.z, s1.z, s0.z); v.w = fmaf(v.w, s1.w, s0.w); }
            b[m] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x4 acc[RW][NTC];
    #pragma unroll
    for (int m = 0; m < RW; ++m)
        #pragma unroll
        for (int q = 0; q < NTC; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bcur[RW], bnxt[RW];
    load(0, bcur);
    __syncthreads();                                              // the weight image is staged
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) load(c + 1, bnxt);
        float4 af[NTC];
        #pragma unroll
        for (int q = 0; q < NTC; ++q) af[q] = amx_ld4(s_w + (size_t)(16 * q + p) * WP + 4 * (4 * c + g));
        #define UC_MFMA(C)                                                                            \
            _Pragma("unroll") for (int m = 0; m < RW; ++m)                                            \
                _Pragma("unroll") for (int q = 0; q < NTC; ++q)                                       \
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q].C, bcur[m].C, acc[m][q], 0, 0, 0);
        UC_MFMA(x) UC_MFMA(y) UC_MFMA(z) UC_MFMA(w)
        #undef UC_MFMA
        #pragma unroll
        for (int m = 0; m < RW; ++m) bcur[m] = bnxt[m];
    }
    // D'[row = cout 16 q + 4 g + r][col = pixel p]: four consecutive couts of one pixel per lane -> one b128 store
    #pragma unroll
    for (int q = 0; q < NTC; ++q) {
        const int co = 16 * q + 4 * g;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) {
            b4.x = co + 0 < a.Cout ? a.bias[co + 0] : 0.f; b4.y = co + 1 < a.Cout ? a.bias[co + 1] : 0.f;
            b4.z = co + 2 < a.Cout ? a.bias[co + 2] : 0.f; b4.w = co + 3 < a.Cout ? a.bias[co + 3] : 0.f;
        }
        #pragma unroll
        for (int m = 0; m < RW; ++m)
            amx_st4(s_v + (size_t)((RW * wave + m) * UC_COLS + p) * VP + co,
                    make_float4(acc[m][q][0] + b4.x, acc[m][q][1] + b4.y, acc[m][q][2] + b4.z, acc[m][q][3] + b4.w));
    }
    __syncthreads();
    // ---- x2 interpolation of the tile: (2 TY) x (2 TX) outputs x CGo channel groups.  A thread owns one (output column,
    // channel group) pair: the horizontal combinations h[r] = wx0 * v[r][c0] + wx1 * v[r][c0 + 1] of the 8 halo rows are
    // formed once and shared by the 12 output rows  o = wy0 * h[r0] + wy1 * h[r0 + 1]  — upsample_fwd_kernel's expression
    // (rows first, then columns) with the common sub-expressions kept in registers; no division inside the row loop.
    const int CGo = a.Cs_out >> 2, H2 = 2 * a.h, W2 = 2 * a.w_;
    for (int j = tid; j < 2 * UC_TX * CGo; j += 256) {
        const int X = j / CGo, cg = j - X * CGo;
        const int GX = 2 * tx0 + X;
        if (GX >= W2) continue;
        const int kx = X >> 1, ox = X & 1;
        float* yb = a.y + (((size_t)n * H2 + 2 * ty0) * W2 + GX) * a.Cs_out + 4 * cg;
        if (a.mode == 1) {                                        // nearest: floor(. / 2)
            #pragma unroll
            for (int Y = 0; Y < 2 * UC_TY; ++Y)
                if (2 * ty0 + Y < H2)
                    amx_st4(yb + (size_t)Y * W2 * a.Cs_out, amx_ld4(s_v + (size_t)(((Y >> 1) + 1) * UC_COLS + kx + 1) * VP + 4 * cg));
            continue;
        }
        // halo column of the first tap: even output 2k -> (k - 1, k) with (.25, .75); odd -> (k, k + 1) with (.75, .25)
        const int c0 = kx + ox;                                   // (interior index k is halo index k + 1)
        const float wx0 = ox ? 0.75f : 0.25f, wx1 = ox ? 0.25f : 0.75f;
        auto hrow = [&](int r) {
            const float4 v0 = amx_ld4(s_v + (size_t)(r * UC_COLS + c0) * VP + 4 * cg);
            const float4 v1 = amx_ld4(s_v + (size_t)(r * UC_COLS + c0 + 1) * VP + 4 * cg);
            return make_float4(wx0 * v0.x + wx1 * v1.x, wx0 * v0.y + wx1 * v1.y, wx0 * v0.z + wx1 * v1.z, wx0 * v0.w + wx1 * v1.w);
        };
        // halo rows (k, k + 1) feed output rows 2k - 1 (weights .75, .25) and 2k (.25, .75): a rolling pair of registers
        float4 h0 = hrow(0);
        #pragma unroll
        for (int k = 0; k <= UC_TY; ++k) {
            const float4 h1 = hrow(k + 1);
            if (k >= 1 && 2 * ty0 + 2 * k - 1 < H2)
                amx_st4(yb + (size_t)(2 * k - 1) * W2 * a.Cs_out,
                        make_float4(0.75f * h0.x + 0.25f * h1.x, 0.75f * h0.y + 0.25f * h1.y, 0.75f * h0.z + 0.25f * h1.z, 0.75f * h0.w + 0.25f * h1.w));
            if (k < UC_TY && 2 * ty0 + 2 * k < H2)
                amx_st4(yb + (size_t)(2 * k) * W2 * a.Cs_out,
                        make_float4(0.25f * h0.x + 0.75f * h1.x, 0.25f * h0.y + 0.75f * h1.y, 0.25f * h0.z + 0.75f * h1.z, 0.25f * h0.w + 0.75f * h1.w));
            h0 = h1;
        }
    }
    }                                                             // tile loop (the barrier at its top protects s_v)
}

static size_t upconv_lds(int ntc, int Cs_in, int rw = 2) {
    const int COP = 16 * ntc, KS = amx_round_up(Cs_in, 16);
    return ((size_t)COP * (KS + 4) + (size_t)4 * rw * UC_COLS * (COP + 4)) * sizeof(float);
}

extern "C" int amx_upconv1x1_supported(int Cin, int Cs_in, int Cout, int Cs_out) {
    if (Cin <= 0 || Cout <= 0 || (Cs_in & 3) || (Cs_out & 3) || Cs_in < Cin || Cs_out < Cout) return 0;
    const int cop = amx_round_up(Cout, 16);
    if (cop > 64 || cop == 48 || Cs_out > cop) return 0;
    // a partial last 16-channel chunk of 2 or 3 k-groups runs TRANSPOSED in conv_kernel.h (TAIL: k-group <-> channel in
    // group), i.e. in another summation order; whole chunks and a one-group tail sum in the order used here -> only those
    // shapes are taken (every UpsampleBlock of the reference's nets: 128 / 64 / 32 and dilnet's 50 -> 52 channels), the
    // rest stays on the two-kernel path, so the result never depends on which path ran
    if ((Cs_in & 15) != 0 && (Cs_in & 15) != 4) return 0;
    return upconv_lds(cop / 16, Cs_in, cop <= 32 ? 4 : 2) <= 160 * 1024 ? 1 : 0;
}

extern "C" int amx_upconv1x1_fwd(const float* x, const float* sc, const float* sh, const float* w, const float* bias,
                                 float* y, int N, int h, int w_, int Cin, int Cs_in, int Cout, int Cs_out, int mode,
                                 void* stream) {
    if (!x || !w || !y || N <= 0 || h <= 0 || w_ <= 0 || (mode != 0 && mode != 1)) AMX_BADARG(1);
    if ((sc == nullptr) != (sh == nullptr)) AMX_BADARG(2);
    if (!amx_upconv1x1_supported(Cin, Cs_in, Cout, Cs_out)) AMX_BADARG(3);
    UpConvArgs a;
    a.x = x; a.sc = sc; a.sh = sh; a.w = w; a.bias = bias; a.y = y;
    a.N = N; a.h = h; a.w_ = w_; a.Cin = Cin; a.Cs_in = Cs_in; a.Cout = Cout; a.Cs_out = Cs_out; a.mode = mode;
    const int ntc = amx_round_up(Cout, 16) / 16;
    // 14 x 14-pixel tiles (4 halo rows per wave) halve the halo re-reads and the barriers per output byte; with 64 couts
    // their v image would leave one workgroup per CU, so that class keeps 6 x 14
    const int rw = (ntc <= 2 && UPCONV_RW4) ? 4 : 2;
    a.tiles_x = amx_ceil_div(w_, UC_TX); a.tiles_y = amx_ceil_div(h, 4 * rw - 2);
    long blocks = (long)N * a.tiles_x * a.tiles_y;
    if (blocks >= 2147483647L) AMX_BADARG(4);
    const size_t lds = upconv_lds(ntc, Cs_in, rw);
    // persistent workgroups (the weight image is staged once each): as many as are RESIDENT — four per CU by registers,
    // fewer where the LDS images allow fewer (a fifth of a second round of workgroups would run on an empty chip)
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const long cap = per_cu * amx_num_cus();
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
#define UC_GO(NTC_, RW_)                                                                                    \
    do {                                                                                                    \
        AMX_ALLOW_160K_LDS(upconv1x1_fwd_kernel<NTC_, RW_>);                                                \
        AMX_LAUNCH((upconv1x1_fwd_kernel<NTC_, RW_>), dim3((unsigned)blocks), dim3(256), lds, s, a);        \
    } while (0)
    if (ntc == 1) { if (rw == 4) UC_GO(1, 4); else UC_GO(1, 2); }
    else if (ntc == 2) { if (rw == 4) UC_GO(2, 4); else UC_GO(2, 2); }
    else UC_GO(4, 2);
#undef UC_GO
    AMX_CHECK_LAUNCH();
    return 0;
}

// Measured and NOT kept (round 5, profiles/r05_logs/r05_upconv_bwd_rejected.log): the backward data path in one pass —
// amx_upsample2x_bwd's gather of the high-resolution gradient into an 8 x 16 low-res tile in LDS (also written out for the
// weight gradient), multiplied by the LDS-resident W right away (the 1x1 data gradient, bit-identical on the emulator and
// on the MI355X) — is SLOWER than the two kernels it replaces on every UpsampleBlock of the config-2 U-Net (137 vs 91 us,
// 206 vs 155, 337 vs 323): the gather is the expensive part, and behind a barrier in a four-wave workgroup it loses more
// than reading dv back costs.
