// conv_kernel.h — the NHWC direct (im2col-free) 3x3 / dilated-3x3 / 1x1 convolution kernel on fp32 MFMA, fused
// with: BatchNorm-affine-on-load of up to two concatenated sources, bias, LeakyReLU and the per-channel batch
// statistics (sum, M2) of the post-activation tensor.  Instantiated by conv_fwd_1x1.hip / conv_fwd_3x3.hip /
// conv_fwd_dil.hip (three translation units so that hipcc compiles them in parallel), dispatched by conv_fwd.hip.
//
// Replaces, on the hot path of the reference (all ATen calls):
//   nn.Conv2d k3 s1 p=d dil=d (+bias)  -> LeakyReLU -> [BatchNorm2d statistics]      atomai/nets/blocks.py:61-76, 300-318
//   torch.cat([skip, up], dim=1) feeding a conv                                      atomai/nets/fcnn.py:132-138, 223
//   BatchNorm2d normalisation of the *previous* layer (applied here, on load)        atomai/nets/blocks.py:71-75
//   nn.Conv2d 1x1 of UpsampleBlock (evaluated at low resolution, it commutes)        atomai/nets/blocks.py:122-132
// The same kernel is the data-gradient (dgrad) engine: a 3x3 dgrad is a 3x3 forward conv with
// spatially flipped, in/out-transposed weights (pack.hip builds that weight image); its two
// concatenated *outputs* (skip / upsampled halves) are written through `y` + `y1`.
//
// Mapping to CDNA4 (gfx950):
//   * implicit GEMM  M = (4*MTW rows x 16) output pixels per work item, N = 16*NT output channels,
//     K = taps x 16-channel chunks.  One wave owns MTW image rows x NT channel tiles and issues
//     v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles): A[pixel][k], B[k][cout].
//   * LDS operand images are laid out so that every fragment is ONE conflict-free ds_read_b128:
//       input  tile  [kgroup][pixel slot][4 channels]   (plane stride == 0 mod 16 slots)
//       weight tile  [tap][kgroup][cout][4 channels]
//     a lane (p = lane&15, g = lane>>4) reads 4 consecutive channels 4g..4g+3 of pixel/cout p and
//     feeds them to 4 consecutive MFMAs, so each MFMA contracts channels {kk, 4+kk, 8+kk, 12+kk}.
//   * EXACT: the dilation is a template constant (1, 2, 4, 6 — every dilation the reference's nets use), so the tile
//     geometry is compile-time: the 9 tap offsets are ds_read immediates instead of ~36 address registers.
//   * LAT (lattice mode, dilations 2 / 4 / 6): a 3x3 convolution with dilation d only ever combines pixels of the same
//     residue class (y mod d, x mod d), i.e. it is d*d independent PLAIN 3x3 convolutions on the sub-images
//     x[ry::d, rx::d].  A workgroup therefore owns a TH x 16 tile of ONE sub-lattice: its input tile has a halo of one
//     lattice step instead of d pixels ((TH+2) x 18 slots instead of (TH+2d) x (16+2d): 1.4x instead of 3.1-4.4x
//     over-fetch at d = 6), the LDS image is that of the plain kernel (3-4 workgroups per CU instead of 1-2) and the
//     compile-time geometry / launch bounds of the plain classes apply.  NHWC makes the strided gather free: the
//     unit of a global access is one pixel's 64-byte channel group either way.
//   * Measured and rejected in round 2 (profiles/r02_conv_persistent_ab.md): persistent workgroups walking several
//     tiles with cross-tile register prefetch, weight images kept resident in LDS for <= 2-chunk layers, and operand
//     fragments software-pipelined across taps — no gain over one workgroup per tile (the hardware
//     dispatcher hides workgroup turnover), slower where the extra state costs a co-resident workgroup.
//   * global->register prefetch of chunk c+1 is issued before the MFMA phase of chunk c; the BN affine and the zero
//     padding are applied when registers are written to LDS (padding must stay zero AFTER the affine, so it cannot
//     be folded into the weights).
//   * epilogue (per wave, no workgroup barrier): bias + LeakyReLU on the accumulators, (sum, M2 about the wave's own
//     mean) of the wave's MTW x 16 pixel strip by shuffles -> one partial-statistics row per wave (bn.hip merges the
//     rows with Chan's formula in fp64: deterministic, no atomics), transpose through a wave-private LDS region,
//     16-byte NHWC stores (+ optional residual addend).
#pragma once
#ifndef AMX_CONV_UNCOND_LOADS
#define AMX_CONV_UNCOND_LOADS 1      // compile-time A/B switch (tools/build_variant_lib.sh), see issue_loads
#endif
#include "amx_device.h"
#include <cstdlib>

#define TILE 16          // output tile is TILE x TILE pixels
#define KG 4             // k-groups (of 4 channels) per chunk -> 16 channels per chunk

struct ConvFwdArgs {
    const float* x0; const float* sc0; const float* sh0; int C0s;
    const float* x1; const float* sc1; const float* sh1; int C1s;
    float in_slope0, in_slope1;   // LeakyReLU applied AFTER the on-load affine of source 0 / 1 (1.0f == none):
                                  // ResBlock's conv -> BatchNorm -> LeakyReLU order (atomai/nets/blocks.py:205-208)
    const float* wpk;    // [nchunk][taps][KG][cop][4]
    const float* bias;   // [cout] or nullptr
    const float* addend; // optional tensor (same shape as y) added to the first output, or nullptr
    float* y;  int Y0s;  // first  output: stored channels Y0s, receives couts [0, Y0s)
    float* y1; int Y1s;  // second output (dgrad of a concat) receives couts [Y0s, Y0s+Y1s), or nullptr
    float* stats;        // [tiles][2][cop] (sum, M2) or nullptr
    // ---- fused classification head (HEAD instantiations, eval mode): the layer's own eval-mode BatchNorm affine and
    // the final 1x1 convolution to K classes are folded into hw / hb on the host; the activation itself is not stored
    const float* hw;     // [hK][cop]: Wpx[k][c] * scale[c]  (0 beyond cout)
    const float* hb;     // [hK]:     bpx[k] + sum_c Wpx[k][c] * shift[c]
    float* hout;         // hmode 0: logits NCHW [N][K][H][W]; 1: probabilities NHWC [N][H][W][K] (sigmoid / softmax)
    int hK, hmode;
    // ---- fused DilatedBlock sum (EPI == 2, eval mode; atomai/nets/blocks.py:321-329): this layer is the block's last
    // one; y receives sum over the block's layers l of [pre_l + a_l + bn_l(a_l)] (pre_l = inverse LeakyReLU of a_l)
    // instead of this layer's activation; ds_a: the earlier layers' activations (same shape as y), ds_sc / ds_sh:
    // eval-mode BatchNorm affines of ALL layers, this layer's last ([nds + 1] vectors of Y0s floats, zeros = no BN)
    const float* ds_a[3];
    const float* ds_sc[4]; const float* ds_sh[4];
    int nds; float ds_inv_slope;
    // ---- data gradient with the producing layer's BatchNorm / LeakyReLU backward formed while loading (conv_ws.hip,
    // round 4): x0 = dy, the gradient w.r.t. the layer's OUTPUT; the convolution input is
    //   dpre = lrelu'(a) * (k1 * dy + k2 * a + k3)       (amx_bn_bwd_apply's arithmetic, never written to HBM)
    const float* bw_aux;                   // the layer's saved activation a (shape of x0), or nullptr
    const float* bw_k1; const float* bw_k2; const float* bw_k3;      // per-channel constants (all three or none)
    float bw_slope;
    // ---- BatchNorm-backward sums of the layer that PRODUCED this launch's output position (round 6, conv_ws.hip BWD
    // launches with one output): the data gradient written to y is the complete dy of a conv -> LeakyReLU -> BatchNorm layer
    // whose saved activation is bs_a (shape of y); the epilogue adds up (sum dy, sum dy * a) per channel into
    // bs_part [amx_conv2d_dgrad_bsum_rows][2][Y0s] — what amx_bn_bwd_reduce would compute in a pass of its own over both tensors
    const float* bs_a; float* bs_part;
    unsigned long long* prof;   // AMX_CONV_PROFILE builds: per-wave phase clocks (or nullptr)
    int N, H, W;
    int cout;            // real number of output channels
    int cop;             // cout rounded up to 16
    int nchunk;
    int tail_kg;         // valid 4-channel k-groups in the LAST chunk (1..KG; KG == the chunk is full)
    int dil;             // dilation (== halo) for 9 taps; ignored for 1 tap
    float slope;         // LeakyReLU negative slope; 1.0f == no activation
    int tiles_x, tiles_y;
    int th;              // tile height in pixels (16 or 32)
    int xcd;             // 1: XCD-aware tile order (neighbouring tiles share an L2)
    // lattice mode, x-packed (round 6): the LAT sub-images of one residue row ry lie SIDE BY SIDE on one tile axis, stride
    // xpack = widest sub-image + 1 (one never-stored gap column between neighbours keeps their halos apart), so a tile
    // column no longer ends at every sub-image: dilation 6 at 512 columns = 6 x 86 -> 33 instead of 6 x 6 = 36 tile columns
    // (12.5 % -> 3 % of the tiles' MFMA work was ragged padding).  0 = one sub-image per tile axis.  xmagic = ceil(2^32 / xpack).
    int xpack; unsigned xmagic;
};

static __device__ __forceinline__ unsigned amx_umulhi(unsigned a, unsigned b) {
    return (unsigned)(((unsigned long long)a * b) >> 32);
}
// packed lattice coordinate u -> image column (or W = "not a pixel": gap column / beyond the last sub-image)
template <int LS>
static __device__ __forceinline__ int amx_xpack_col(int u, int xpack, unsigned xmagic, int W) {
    if (u < 0) return -1;
    const int q = (int)amx_umulhi((unsigned)u, xmagic), l = u - q * xpack;
    return (q < LS && l < (W - q + LS - 1) / LS) ? q + l * LS : W;
}

// TAIL: compiled-in support for a partial last chunk (compute_tail).  It is a separate instantiation because the
// extra unrolled tap loop costs code and registers that the layers with channel counts in multiples of 16 need not pay.
//
// Compile-time experiment switches (tools/build_variant_lib.sh builds one library per combination for in-process A/B):
//   AMX_CONV_EXACT  0: instantiations marked EXACT still read the dilation from the arguments (round-1 behaviour)
#ifndef AMX_CONV_EXACT
#define AMX_CONV_EXACT 1
#endif
// AMX_CONV_GLDS 1: the weight image of a chunk goes global -> LDS by LDS-DMA (global_load_lds_dwordx4) instead of
// through WLD float4 registers per thread (-20 VGPRs on the 32-cout classes, no ds_write pass); the DMA of chunk c+1 can
// only be issued after the barrier that ends chunk c (the weight image is single-buffered), so its latency is exposed.
// Measured (profiles/r02_logs/r02y_probe.log, r02y_step_ab.log): at 4 waves/SIMD +1..5 % on the <= 32-channel layers, -1.5 % on
// the 128-channel ones, 19.28 -> 19.35 ms in the step; asking for 5 waves/SIMD (AMX_CONV_GLDS_WAVES) spills 15 registers
// on the 32-cout class and is 3-10 % slower, and the 16-cout class, which fits 5 waves without spilling, gains nothing.
// Off by default.
#ifndef AMX_CONV_GLDS
#define AMX_CONV_GLDS 0
#endif
// AMX_CONV_STAGE_MAP (experiment switch, round 6): which (pixel slot, 4-channel group) a thread loads and stages.
//   0: channel group fastest (tid & 3 = group, tid >> 2 = slot): 4 consecutive lanes read one pixel's 64 contiguous bytes; the
//      ds_write_b128 of 8 consecutive lanes then lands on 2 slots x 4 planes whose strides are == 0 mod 32 banks — a 4-way
//      conflict (30.5 instead of 8 LDS cycles per wave instruction, profiles/r06_lds_conflicts.md `stage_write<192, 0>`);
//   1: 8 consecutive lanes take 8 consecutive slots of ONE group (conflict-free store; the wave still covers 16 pixels x 4
//      groups = the same 16 cache lines per load instruction).
#ifndef AMX_CONV_STAGE_MAP
#define AMX_CONV_STAGE_MAP 0
#endif
#ifndef AMX_CONV_EPI_PAD
#define AMX_CONV_EPI_PAD 0          // floats of padding per pixel row of the epilogue's transposition buffer; 4 removes the
                                    // 4-way bank conflict of the scalar stores but measured SLOWER in the step (18.56 ->
                                    // 19.01 ms, profiles/r03_conv_epi_pad_ab.log): the b128 read-back then straddles rows
#endif
// AMX_CONV_PRIO_OUT / AMX_CONV_PRIO_MFMA: s_setprio of a wave outside / inside its MFMA phases (experiment switches; equal
// values = no s_setprio at all).  The issue arbiter serves the oldest ready wave first, and a wave whose next MFMA waits for
// the busy matrix pipe stands in front of the other waves' prologue / epilogue instructions (conv_ws.hip found its
// producer waves starved that way).  Measured here (profiles/r03_conv_prio_ab.log): OUT = 3 18.54 -> 18.73 ms per U-Net
// step and 1.312 -> 1.406 ms per dilnet frame, OUT = 1 the same, MFMA = 3 18.58 / 1.427: with four symmetric waves per
// SIMD the default oldest-first order is the best one — off.
#ifndef AMX_CONV_PRIO_OUT
#define AMX_CONV_PRIO_OUT 0
#endif
#ifndef AMX_CONV_PRIO_MFMA
#define AMX_CONV_PRIO_MFMA 0
#endif
#if !defined(AMX_EMU) && (AMX_CONV_PRIO_OUT != AMX_CONV_PRIO_MFMA)
#define AMX_SETPRIO(v) __builtin_amdgcn_s_setprio(v)
#else
#define AMX_SETPRIO(v) do { } while (0)
#endif
#ifndef AMX_CONV_GLDS_WAVES
#define AMX_CONV_GLDS_WAVES 5       // waves per SIMD requested for the 8-row thin classes when the weights go by LDS-DMA
#endif
#if defined(AMX_EMU)
#undef AMX_CONV_GLDS
#define AMX_CONV_GLDS 0
#endif
// AMX_CONV_PROFILE (dev builds only, tools/gpu_conv_phases.py): every wave records the shader clock at its phase
// boundaries into a.prof: [workgroup][wave][16] 64-bit ticks.
#ifdef AMX_CONV_PROFILE
#define AMX_TICK(slot)                                                                                          \
    do {                                                                                                        \
        if (a.prof && lane == 0)                                                                                \
            a.prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + (slot)] = \
                __builtin_amdgcn_s_memtime();                                                                   \
    } while (0)
#else
#define AMX_TICK(slot) do { } while (0)
#endif
// Waves per SIMD the plain-3x3 instantiations are built for.  Co-resident waves are what keeps the matrix pipe fed
// (utilisation ~ N * t_mfma / (t_mfma + t_other) per SIMD, profiles/r02_conv_phases.md), and these variants sit a
// handful of registers above the next allocation step: the bound makes the allocator take the step.
#ifndef AMX_CONV_REM_GLDS
#define AMX_CONV_REM_GLDS 1
#endif
#ifndef AMX_CONV_REM_WAVES
#define AMX_CONV_REM_WAVES 3        // waves per SIMD the remainder-column classes are built for (experiment switch)
#endif
template <int TAPS, int NT, int MAXHALO, int MTW, bool TAIL, int REM = 0>
struct ConvWaves {
    static constexpr int value = REM ? AMX_CONV_REM_WAVES
                                     : (TAPS == 9 && MAXHALO == 1)
                                     ? ((MTW == 2 && NT <= 2) ? (AMX_CONV_GLDS ? AMX_CONV_GLDS_WAVES : 4) : ((MTW == 4 && NT == 1) ? 3 : 1))
                                     : 1;
};

// REM (round 4): 1..3 extra 4-wide column blocks on v_mfma_f32_4x4x1_16b_f32 behind the NT 16-wide tiles, so that a
// workgroup covers EXACTLY the stored channels of a layer whose width is not a multiple of 16 — dilnet's 25 / 50 filters
// (28 / 52 stored): 16 + 3 x 4 and 3 x 16 + 4 columns in ONE cout block instead of 32 and 2 x 32 (27.7 % of the MFMA work
// of a dilnet forward was column padding, profiles/r03_pmc_dilnet.md, and the second cout block staged the input tile a
// second time).  The 4x4x1 form (16 independent 4x4 blocks, K = 1; lane map measured in tools/micro/mfma4x4_probe.hip:
// block = lane / 4, A row = B column = lane % 4, D register = row) runs at the rate of the 16x16x4 form — 8 cycles for a
// quarter of the MACs — and takes the SAME A fragment registers: lane (g, p) holds channels 4g..4g+3 of pixel p, so block
// 4g + p/4 multiplies pixels 4(p/4)..+3 at channel 4g + c with the weights of couts n0R + p%4; the four channel sets g
// of a pixel are separate blocks and are added once, by two xor-shuffles, in the epilogue.
template <int TAPS, int NT, int MAXHALO, bool EXACT, int MTW, int EPI, bool TAIL = false, int LAT = 0, int REM = 0>
__global__ __launch_bounds__(256, (ConvWaves<TAPS, NT, MAXHALO, MTW, TAIL, REM>::value)) void conv_fwd_kernel(ConvFwdArgs a) {
    static_assert(LAT == 0 || (TAPS == 9 && MAXHALO == 1 && EXACT), "lattice mode runs the plain 3x3 geometry");
    // EPI: 0 = store the activation; 1 = classification head fused in (eval); 2 = sum of a DilatedBlock fused in (eval)
    constexpr bool HEAD = EPI == 1, DSUM = EPI == 2;
    constexpr int TH = 4 * MTW;                                  // tile rows (MTW image rows per wave)
    constexpr int NB = NT * 16 + REM * 4;                        // columns (couts) of this workgroup
    constexpr int RQ = REM ? REM : 1;
    constexpr int MAXI = TILE + 2 * MAXHALO;
    constexpr int XLD = ((TH + 2 * MAXHALO) * MAXI * KG + 255) / 256;          // float4 loads per thread (input)
    constexpr int WLD = (TAPS * KG * NB + 255) / 256;            // float4 loads per thread (weights)
    // weights by LDS-DMA instead of through registers: everywhere under AMX_CONV_GLDS (experiment), and on the
    // remainder-column classes, whose 52-column weight image would hold 32 registers per thread across the MFMA phase
#ifdef AMX_EMU
    constexpr bool GLDS = false;
#else
    constexpr bool GLDS = AMX_CONV_GLDS || (REM && AMX_CONV_REM_GLDS);
#endif
    AMX_DYN_SMEM(float, smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    AMX_SETPRIO(AMX_CONV_PRIO_OUT);
    AMX_TICK(0);
    // EXACT: the dilation equals MAXHALO (1, 2, 4, 6 — every dilation the reference's nets use) and the whole tile
    // geometry is compile-time: the slot index math of the loaders divides by constants instead of running a ~40
    // instruction integer division per slot, and the 9 tap offsets of the fragment reads are immediates.
    const int halo = (TAPS == 9) ? ((AMX_CONV_EXACT && EXACT) ? MAXHALO : a.dil) : 0;
    const int IW = TILE + 2 * halo, IH = TH + 2 * halo;
    const int plane = amx_round_up(IH * IW, 16);                 // slots (16 B) per k-group plane
    // one stage = input image [KG][plane][4] + weight image [TAPS][KG][NB][4]
    const int stage_floats = KG * plane * 4 + TAPS * KG * NB * 4;
    float* s_red = smem;                                         // reused after the K loop

    // logical (tile, cout-block) of this workgroup: cout blocks of one tile adjacent, tiles in raster order,
    // contiguous ranges of them per XCD
    unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (a.xcd) lin = amx_xcd_remap(lin, gridDim.x * gridDim.y);
    int t = a.xcd ? (int)(lin / gridDim.y) : (int)blockIdx.x;
    const int ob = a.xcd ? (int)(lin % gridDim.y) : (int)blockIdx.y;
    // lattice mode: the d*d residue classes of one tile position are adjacent workgroups (they share cache lines);
    // all tile coordinates below are then LATTICE coordinates: image (y, x) = (ry + ly*LAT, rx + lx*LAT)
    constexpr int LS = LAT ? LAT : 1;
    int ry = 0, rx = 0;
    const int xp = LAT ? a.xpack : 0;                             // x-packed lattice tiles (ConvFwdArgs::xpack)
    if (LAT) {
        if (xp) { ry = t % LS; t /= LS; }
        else { const int rr = t % (LS * LS); t /= (LS * LS); ry = rr / LS; rx = rr - ry * LS; }
    }
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; const int n = t / a.tiles_y;
    const int n0 = ob * NB;                                      // first cout of this workgroup
    const int gy0 = ty * TH - halo, gx0 = tx * TILE - halo;

    AMX_TICK(14);
    // ---- per-thread load descriptors (constant over chunks: kg = tid&3) ----
    const int my_kg = AMX_CONV_STAGE_MAP ? ((tid >> 3) & (KG - 1)) : (tid & (KG - 1));
    auto slot_of = [&](int i) { return AMX_CONV_STAGE_MAP ? ((tid & 7) + 8 * (tid >> 5) + 64 * i) : ((tid + i * 256) >> 2); };
    const int nslots = IH * IW;
    constexpr bool UNCOND = AMX_CONV_UNCOND_LOADS && NT <= 2;
    int x_off[XLD];                                              // pixel offset ((n*H+y)*W+x) or -1
    #pragma unroll
    for (int i = 0; i < XLD; ++i) {
        const int pix = slot_of(i);
        int off = -1;
        if (pix < nslots) {
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = ry + (gy0 + iy) * LS;                                // (< 0 exactly when the lattice index is)
            const int gx = (LAT && xp) ? amx_xpack_col<LS>(gx0 + ix, xp, a.xmagic, a.W) : rx + (gx0 + ix) * LS;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) off = (n * a.H + gy) * a.W + gx;
        }
        x_off[i] = off;
    }
    AMX_TICK(15);

    float4 xr[XLD];
    float4 wr[GLDS ? 1 : WLD];
    float4 r_sc, r_sh;
    float r_islope = 1.f;                                        // post-affine LeakyReLU slope of this thread's source

    auto issue_loads = [&](int chunk) {
        const int ch = (chunk * KG + my_kg) * 4;                 // channel in the concatenated space
        const float* src = nullptr; const float* sc = nullptr; const float* sh = nullptr;
        int Cs = 0, c = 0;
        if (ch < a.C0s) { src = a.x0; sc = a.sc0; sh = a.sh0; Cs = a.C0s; c = ch; }
        else if (ch - a.C0s < a.C1s) { src = a.x1; sc = a.sc1; sh = a.sh1; Cs = a.C1s; c = ch - a.C0s; }
        r_sc = make_float4(1.f, 1.f, 1.f, 1.f);
        r_sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src && sc) { r_sc = amx_ld4(sc + c); r_sh = amx_ld4(sh + c); }
        r_islope = src == a.x1 ? a.in_slope1 : a.in_slope0;
        // The loads of a chunk are UNCONDITIONAL per slot (UNCOND; not on the 64-column classes, where the one extra register
        // it costs halves the occupancy): a slot outside the image reads pixel 0 of
        // its source and is zeroed when the chunk is staged.  The predicated form compiled to an exec-mask branch, four
        // zero-initialising moves and a 64-bit multiply-add pair around every load (~12 instructions per load in a kernel
        // that is issue-bound on the thin layers).
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            if (UNCOND) {
                xr[i] = src ? amx_ld4(src + (size_t)(x_off[i] >= 0 ? x_off[i] : 0) * Cs + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src && x_off[i] >= 0) xr[i] = amx_ld4(src + (size_t)x_off[i] * Cs + c);
            }
        }
        if constexpr (!GLDS) {
            const float* wsrc = a.wpk + (size_t)chunk * TAPS * KG * a.cop * 4;
            #pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int idx = tid + i * 256;                   // over [TAPS*KG][NB]
                wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < TAPS * KG * NB) {
                    const int row = idx / NB, col = idx - row * NB;
                    if (n0 + col < a.cop) wr[i] = amx_ld4(wsrc + ((size_t)row * a.cop + n0 + col) * 4);
                }
            }
        }
    };
#ifndef AMX_EMU
    // weight image of a chunk by LDS-DMA: one wave instruction moves 1 KiB = 64 / NB rows of the [TAPS*KG][NB][4] image
    // (LDS destination = wave-uniform base + lane * 16 B, the global source is per lane)
    auto dma_weights = [&](int chunk) {
        const float* wsrc = a.wpk + (size_t)chunk * TAPS * KG * a.cop * 4;
        float* s_w = smem + KG * plane * 4;
        constexpr int NI = (TAPS * KG * NB + 63) / 64;           // wave instructions per chunk
        #pragma unroll
        for (int i = 0; i < (NI + 3) / 4; ++i) {
            const int ii = wave + 4 * i;
            if (ii < NI) {
                const int idx = ii * 64 + lane;
                const int row = idx / NB, col = idx - row * NB;
                if (idx < TAPS * KG * NB && n0 + col < a.cop)
                    __builtin_amdgcn_global_load_lds(wsrc + ((size_t)row * a.cop + n0 + col) * 4, s_w + (size_t)ii * 256, 16, 0, 0);
            }
        }
    };
#endif

    auto stage_to_lds = [&](int stage, bool tchunk) {
        float* s_in = smem + (size_t)stage * stage_floats;
        float* s_w = s_in + KG * plane * 4;
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int pix = slot_of(i);
            if (pix < nslots) {
                float4 v = xr[i];
                if (UNCOND && x_off[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x_off[i] >= 0) {                             // padding stays exactly zero
                    v.x = fmaf(v.x, r_sc.x, r_sh.x); v.y = fmaf(v.y, r_sc.y, r_sh.y);
                    v.z = fmaf(v.z, r_sc.z, r_sh.z); v.w = fmaf(v.w, r_sc.w, r_sh.w);
                    if (r_islope != 1.f) {
                        v.x = v.x > 0.f ? v.x : v.x * r_islope; v.y = v.y > 0.f ? v.y : v.y * r_islope;
                        v.z = v.z > 0.f ? v.z : v.z * r_islope; v.w = v.w > 0.f ? v.w : v.w * r_islope;
                    }
                }
                if (TAIL && tchunk) {
                    // partial last chunk: stored transposed — plane e holds channel e of every k-group, this thread's
                    // k-group in component my_kg (the weight image of that chunk is packed the same way, pack.hip)
                    if (my_kg < a.tail_kg) {
                        s_in[((size_t)0 * plane + pix) * 4 + my_kg] = v.x;
                        s_in[((size_t)1 * plane + pix) * 4 + my_kg] = v.y;
                        s_in[((size_t)2 * plane + pix) * 4 + my_kg] = v.z;
                        s_in[((size_t)3 * plane + pix) * 4 + my_kg] = v.w;
                    }
                } else {
                    amx_st4(s_in + ((size_t)my_kg * plane + pix) * 4, v);
                }
            }
        }
        if constexpr (!GLDS) {
            #pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int idx = tid + i * 256;
                if (idx < TAPS * KG * NB) amx_st4(s_w + (size_t)idx * 4, wr[i]);
            }
        }
    };

    f32x4 acc[MTW][NT];
    f32x4 accr[MTW][RQ];                                         // remainder blocks (REM): partial over this lane's channel set g
    #pragma unroll
    for (int m = 0; m < MTW; ++m) {
        #pragma unroll
        for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        #pragma unroll
        for (int q = 0; q < RQ; ++q) accr[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    auto compute_taps = [&](int stage, int tap0, int tap1) {
        const float* s_in = smem + (size_t)stage * stage_floats;
        const float* s_w = s_in + KG * plane * 4;
        #pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap < tap0 || tap >= tap1) continue;
            const int dy = (TAPS == 9) ? (tap / 3 - 1) * halo : 0;
            const int dx = (TAPS == 9) ? (tap % 3 - 1) * halo : 0;
            float4 af[MTW], bf[NT];
            #pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int slot = (wave * MTW + m + halo + dy) * IW + (p + halo + dx);
                af[m] = amx_ld4(s_in + ((size_t)g * plane + slot) * 4);
            }
            #pragma unroll
            for (int q = 0; q < NT; ++q)
                bf[q] = amx_ld4(s_w + ((size_t)(tap * KG + g) * NB + q * 16 + p) * 4);
            float4 br[RQ];
            if (REM) {
                #pragma unroll
                for (int q = 0; q < REM; ++q)
                    br[q] = amx_ld4(s_w + ((size_t)(tap * KG + g) * NB + NT * 16 + q * 4 + (p & 3)) * 4);
            }
            // k-subgroup outermost: consecutive MFMAs hit DIFFERENT accumulators (the 16x16x4 f32 MFMA has a
            // 40-cycle dependent latency vs a 32-cycle issue interval)
            #define AMX_CONV_MFMA(C)                                                                    \
                _Pragma("unroll") for (int m = 0; m < MTW; ++m) {                                       \
                    _Pragma("unroll") for (int q = 0; q < NT; ++q)                                      \
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].C, bf[q].C, acc[m][q], 0, 0, 0); \
                    if (REM) {                                                                          \
                        _Pragma("unroll") for (int q = 0; q < REM; ++q)                                 \
                            accr[m][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[m].C, br[q].C, accr[m][q], 0, 0, 0); \
                    }                                                                                   \
                }
            AMX_CONV_MFMA(x) AMX_CONV_MFMA(y) AMX_CONV_MFMA(z) AMX_CONV_MFMA(w)
            #undef AMX_CONV_MFMA
        }
    };

    // Last chunk with fewer than KG valid k-groups (channel counts that are not multiples of 16, e.g. dilnet's
    // 25 / 50 filters -> 28 / 52 stored channels).  That chunk sits TRANSPOSED in LDS (stage_to_lds / pack.hip): the
    // ordinary fragment read returns channel g of k-groups 0..3 in components x..w, so one MFMA per VALID k-group
    // contracts that group's 4 channels — nkg MFMAs per tap and tile instead of four, with the fragment-read code
    // (and its registers) of the main path.
    auto compute_tail = [&](int nkg) {
        const float* s_in = smem;
        const float* s_w = s_in + KG * plane * 4;
        #pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? (tap / 3 - 1) * halo : 0;
            const int dx = (TAPS == 9) ? (tap % 3 - 1) * halo : 0;
            float4 af[MTW], bf[NT];
            #pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int slot = (wave * MTW + m + halo + dy) * IW + (p + halo + dx);
                af[m] = amx_ld4(s_in + ((size_t)g * plane + slot) * 4);
            }
            #pragma unroll
            for (int q = 0; q < NT; ++q)
                bf[q] = amx_ld4(s_w + ((size_t)(tap * KG + g) * NB + q * 16 + p) * 4);
            float4 br[RQ];
            if (REM) {
                #pragma unroll
                for (int q = 0; q < REM; ++q)
                    br[q] = amx_ld4(s_w + ((size_t)(tap * KG + g) * NB + NT * 16 + q * 4 + (p & 3)) * 4);
            }
            #define AMX_CONV_MFMA(C)                                                                    \
                _Pragma("unroll") for (int m = 0; m < MTW; ++m) {                                       \
                    _Pragma("unroll") for (int q = 0; q < NT; ++q)                                      \
                        acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].C, bf[q].C, acc[m][q], 0, 0, 0); \
                    if (REM) {                                                                          \
                        _Pragma("unroll") for (int q = 0; q < REM; ++q)                                 \
                            accr[m][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[m].C, br[q].C, accr[m][q], 0, 0, 0); \
                    }                                                                                   \
                }
            AMX_CONV_MFMA(x)
            if (nkg > 1) { AMX_CONV_MFMA(y) }
            if (nkg > 2) { AMX_CONV_MFMA(z) }
            #undef AMX_CONV_MFMA
        }
    };

    // the bias is fetched up front: a global load at the head of the epilogue would expose a full memory round trip
    // (not for the widest tile: 4 more live registers would push it past 256 = one workgroup per CU instead of two)
    constexpr bool PREB = !(NT == 4 && MTW == 4) && !TAIL;      // (TAIL classes: the registers buy a 4th wave per SIMD)
    float bias_q[NT];
    #pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int co = n0 + q * 16 + p;
        bias_q[q] = (PREB && a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    float bias_r[RQ];
    #pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int co = n0 + NT * 16 + q * 4 + (p & 3);
        bias_r[q] = (REM && a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    issue_loads(0);
#ifndef AMX_EMU
    if constexpr (GLDS) dma_weights(0);
#endif
    AMX_TICK(1);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        stage_to_lds(0, TAIL && chunk + 1 == a.nchunk && a.tail_kg < KG);
        if (chunk < 2) AMX_TICK(2 + 5 * chunk);
        __syncthreads();
        if (chunk < 2) AMX_TICK(3 + 5 * chunk);
        if (chunk + 1 < a.nchunk) issue_loads(chunk + 1);
        if (chunk < 2) AMX_TICK(4 + 5 * chunk);
        AMX_SETPRIO(AMX_CONV_PRIO_MFMA);
        if (TAIL && chunk + 1 == a.nchunk && a.tail_kg < KG) compute_tail(a.tail_kg);
        else compute_taps(0, 0, TAPS);
        AMX_SETPRIO(AMX_CONV_PRIO_OUT);
        if (chunk < 2) AMX_TICK(5 + 5 * chunk);
        __syncthreads();
#ifndef AMX_EMU
        if constexpr (GLDS) { if (chunk + 1 < a.nchunk) dma_weights(chunk + 1); }     // (after the barrier: every wave is done with chunk's image)
#endif
        if (chunk < 2) AMX_TICK(6 + 5 * chunk);
    }

    // ---- epilogue: bias + LeakyReLU in registers, per-WAVE statistics (shuffles only: no LDS, no
    // workgroup barrier), then the accumulators are transposed through a wave-private LDS region and leave as 16-byte
    // stores along the channel axis.  Measured with per-phase shader-clock stamps (tools/gpu_conv_phases.py): the 16
    // scalar stores per lane of the direct form cost ~600 cycles EACH under load (the vector-memory issue path is the
    // contended resource, not bandwidth) and the three barriers of the workgroup-level statistics another ~5 k cycles:
    // together a third of a wave's lifetime on the <= 32-channel layers.
    // C/D fragment: column (cout) = lane&15 = p, row (pixel x) = 4*g + reg.
    const int oy0 = ty * TH + wave * MTW, ox0 = tx * TILE + 4 * g;
    const int ctot = a.Y0s + a.Y1s;
    float lsum[NT];
    #pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int co = n0 + q * 16 + p;
        const float b = PREB ? bias_q[q] : ((a.bias && co < a.cout) ? a.bias[co] : 0.f);
        lsum[q] = 0.f;
        #pragma unroll
        for (int m = 0; m < MTW; ++m)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][q][r] + b;
                v = v > 0.f ? v : v * a.slope;
                const bool ok = (ry + (oy0 + m) * LS < a.H) && ((LAT && xp) || rx + (ox0 + r) * LS < a.W) && co < ctot;
                v = ok ? v : 0.f;
                lsum[q] += v;
                acc[m][q][r] = v;
            }
    }
    // remainder blocks: add the four channel sets (lanes g = 0..3 of a pixel group), then the same bias / activation; after
    // the butterfly every g holds the full value of pixel x = 4 * (p >> 2) + reg, cout n0 + NT * 16 + 4 q + (p & 3)
    float lsum_r[RQ];
    if (REM) {
        #pragma unroll
        for (int q = 0; q < REM; ++q) {
            const int co = n0 + NT * 16 + q * 4 + (p & 3);
            lsum_r[q] = 0.f;
            #pragma unroll
            for (int m = 0; m < MTW; ++m)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = accr[m][q][r];
                    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                    v += bias_r[q];
                    v = v > 0.f ? v : v * a.slope;
                    const bool ok = (ry + (oy0 + m) * LS < a.H) && ((LAT && xp) || rx + (tx * TILE + 4 * (p >> 2) + r) * LS < a.W) && co < ctot;
                    v = ok ? v : 0.f;
                    lsum_r[q] += v;
                    accr[m][q][r] = v;
                }
        }
    }
    AMX_TICK(12);
    // sub-image extent of this workgroup's residue class (the image itself outside lattice mode)
    const int Hs = LAT ? (a.H - ry + LS - 1) / LS : a.H, Ws = LAT ? (a.W - rx + LS - 1) / LS : a.W;
    if (a.stats && oy0 < (LAT ? (a.H + LS - 1) / LS : a.H)) {
        // one statistics row per wave: (sum, M2 about the wave's own mean) over its MTW x 16 pixel strip; the host
        // sees MTW as the "tile height" (amx_conv2d_tile_h) and the merge kernels weight rows by their pixel counts
        // (lattice mode: rows [n][ry][rx][strip][tx], strips of residue classes with a shorter sub-image may be empty)
        const int vy = max(0, min(MTW, Hs - oy0)), vx = max(0, min(TILE, Ws - tx * TILE));
        const float inv_cnt = vy * vx > 0 ? 1.0f / (float)(vy * vx) : 0.f;
        const int sub_y = ty * 4 + wave, subs_y = amx_ceil_div(LAT ? (a.H + LS - 1) / LS : a.H, MTW);
        const size_t row = ((size_t)((n * LS + ry) * LS + rx) * subs_y + sub_y) * a.tiles_x + tx;
        #pragma unroll
        for (int q = 0; q < NT; ++q) {
            float sm = lsum[q];
            sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
            const float mu = sm * inv_cnt;
            float s2 = 0.f;
            #pragma unroll
            for (int m = 0; m < MTW; ++m)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (oy0 + m < Hs) && (ox0 + r < Ws);
                    const float d = acc[m][q][r] - mu;
                    s2 += ok ? d * d : 0.f;
                }
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            const int co = n0 + q * 16 + p;
            if (g == 0 && co < a.cop) {
                a.stats[(row * 2) * a.cop + co] = sm;
                a.stats[(row * 2 + 1) * a.cop + co] = s2;
            }
        }
        if (REM) {
            // a lane holds 4 pixels x MTW rows of one remainder cout; the strip's 16 columns are the 4 lane groups p >> 2
            #pragma unroll
            for (int q = 0; q < REM; ++q) {
                float sm = lsum_r[q];
                sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 8);
                const float mu = sm * inv_cnt;
                float s2 = 0.f;
                #pragma unroll
                for (int m = 0; m < MTW; ++m)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (oy0 + m < Hs) && (tx * TILE + 4 * (p >> 2) + r < Ws);
                        const float d = accr[m][q][r] - mu;
                        s2 += ok ? d * d : 0.f;
                    }
                s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);
                const int co = n0 + NT * 16 + q * 4 + (p & 3);
                if (lane < 4 && co < a.cop) {
                    a.stats[(row * 2) * a.cop + co] = sm;
                    a.stats[(row * 2 + 1) * a.cop + co] = s2;
                }
            }
        }
    }
    // transpose: [row m][pixel x][cout] in this wave's LDS region, MH rows at a time
    constexpr int MH = MTW < 2 ? MTW : 2;
    constexpr int CG = NB / 4;                                   // float4 groups per pixel
    // Row stride NB + EPAD floats (experiment switch, default 0): the four lane groups g of a fragment store to pixel rows
    // 4g + r, i.e. 4 * stride floats apart — with a stride of NB (a multiple of 16) all four land on the same 16 banks, a
    // 4-way conflict on every scalar store (SQ_LDS_BANK_CONFLICT 0.28-0.41 of the LDS cycles, profiles/r02_pmc_sq.md);
    // + 4 floats rotates each group by 16 banks — and loses more on the read-back side than it gains (see above).
    constexpr int EPAD = AMX_CONV_EPI_PAD;
    constexpr int NBE = NB + EPAD;
    float* s_epi = smem + (size_t)wave * (MH * TILE * NBE);
    // HEAD: this lane's slice (4 couts) of the folded head weights; lane % CG is the lane's float4 group of a pixel in
    // every iteration of the loops below (64 and MH * TILE * CG are multiples of CG)
    // (HEAD with remainder columns: CG = 7 float4 groups per pixel are dealt to CGH = 8 lanes, the eighth contributes 0)
    constexpr int CGH = CG <= 1 ? 1 : (CG <= 2 ? 2 : (CG <= 4 ? 4 : (CG <= 8 ? 8 : 16)));
    constexpr int CGX = HEAD ? CGH : CG;                         // lanes per pixel in the store loop
    float4 hwq[HEAD ? 3 : 1];
    if (HEAD) {
        #pragma unroll
        for (int k = 0; k < 3; ++k)
            hwq[k] = (k < a.hK && lane % CGH < CG) ? amx_ld4(a.hw + (size_t)k * a.cop + (lane % CGH) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    #pragma unroll
    for (int m0 = 0; m0 < MTW; m0 += MH) {
        #pragma unroll
        for (int mm = 0; mm < MH; ++mm)
            #pragma unroll
            for (int q = 0; q < NT; ++q)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    s_epi[(mm * TILE + 4 * g + r) * NBE + q * 16 + p] = acc[m0 + mm][q][r];
        if (REM && g == 0) {                                     // (every g holds the same values after the butterfly)
            #pragma unroll
            for (int mm = 0; mm < MH; ++mm)
                #pragma unroll
                for (int q = 0; q < REM; ++q)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s_epi[(mm * TILE + 4 * (p >> 2) + r) * NBE + NT * 16 + q * 4 + (p & 3)] = accr[m0 + mm][q][r];
        }
        amx_wave_sync();                                         // wave-private region: no workgroup barrier needed
        #pragma unroll
        for (int it = 0; it < (MH * TILE * CGX + 63) / 64; ++it) {
            const int e = it * 64 + lane;
            if ((MH * TILE * CGX) % 64 && e >= MH * TILE * CGX) continue;    // (REM: CG = 7 / 13 float4 groups per pixel)
            const int pix = e / CGX, cgp = e - pix * CGX;
            const int mm = pix / TILE, x = pix - mm * TILE;
            const int oy = ry + (oy0 + m0 + mm) * LS;
            const int ox = (LAT && xp) ? amx_xpack_col<LS>(tx * TILE + x, xp, a.xmagic, a.W) : rx + (tx * TILE + x) * LS;
            const int co = n0 + cgp * 4;
            if (HEAD) {
                // the final 1x1 convolution (own BatchNorm affine folded in) on the transposed tile: each of the CG
                // lanes of a pixel contracts its 4 couts, a butterfly over those lanes sums them; the activation is
                // not written at all
                const float4 v = cgp < CG ? amx_ld4(s_epi + (size_t)pix * NBE + cgp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                float lg[3];
                #pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float d = v.x * hwq[k].x + v.y * hwq[k].y + v.z * hwq[k].z + v.w * hwq[k].w;
                    #pragma unroll
                    for (int o = 1; o < CGH; o <<= 1) d += __shfl_xor(d, o);
                    lg[k] = d + (k < a.hK ? a.hb[k] : 0.f);
                }
                if (cgp == 0 && oy < a.H && ox < a.W) {
                    const size_t pq = (size_t)(n * a.H + oy) * a.W + ox;
                    if (a.hmode == 0) {
                        for (int k = 0; k < a.hK; ++k) a.hout[((size_t)n * a.hK + k) * a.H * a.W + (size_t)oy * a.W + ox] = lg[k];
                    } else if (a.hK == 1) {
                        a.hout[pq] = 1.f / (1.f + expf(-lg[0]));
                    } else {
                        float mx = lg[0];
                        for (int k = 1; k < a.hK; ++k) mx = fmaxf(mx, lg[k]);
                        float sme = 0.f;
                        for (int k = 0; k < a.hK; ++k) { lg[k] = expf(lg[k] - mx); sme += lg[k]; }
                        for (int k = 0; k < a.hK; ++k) a.hout[pq * a.hK + k] = lg[k] / sme;
                    }
                }
                continue;
            }
            if (DSUM) {
                if (oy < a.H && ox < a.W && co < a.Y0s) {
                    const size_t o = ((size_t)(n * a.H + oy) * a.W + ox) * a.Y0s + co;
                    auto term = [&](float4 t, const float* scp, const float* shp) {
                        const float4 sc = amx_ld4(scp + co), sh = amx_ld4(shp + co);
                        float4 r;
                        r.x = (t.x > 0.f ? t.x : t.x * a.ds_inv_slope) + t.x + fmaf(t.x, sc.x, sh.x);
                        r.y = (t.y > 0.f ? t.y : t.y * a.ds_inv_slope) + t.y + fmaf(t.y, sc.y, sh.y);
                        r.z = (t.z > 0.f ? t.z : t.z * a.ds_inv_slope) + t.z + fmaf(t.z, sc.z, sh.z);
                        r.w = (t.w > 0.f ? t.w : t.w * a.ds_inv_slope) + t.w + fmaf(t.w, sc.w, sh.w);
                        return r;
                    };
                    float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int l = 0; l < a.nds; ++l) {                    // same order as amx_dilated_sum: layers ascending
                        const float4 r = term(amx_ld4(a.ds_a[l] + o), a.ds_sc[l], a.ds_sh[l]);
                        acc4.x += r.x; acc4.y += r.y; acc4.z += r.z; acc4.w += r.w;
                    }
                    const float4 r = term(amx_ld4(s_epi + (size_t)pix * NBE + cgp * 4), a.ds_sc[a.nds], a.ds_sh[a.nds]);
                    acc4.x += r.x; acc4.y += r.y; acc4.z += r.z; acc4.w += r.w;
                    amx_st4(a.y + o, acc4);
                }
                continue;
            }
            if (oy < a.H && ox < a.W && co < ctot) {
                float4 v = amx_ld4(s_epi + (size_t)pix * NBE + cgp * 4);
                float* dst; int Cd, cd;
                if (co < a.Y0s) { dst = a.y; Cd = a.Y0s; cd = co; } else { dst = a.y1; Cd = a.Y1s; cd = co - a.Y0s; }
                const size_t o = ((size_t)(n * a.H + oy) * a.W + ox) * Cd + cd;
                if (a.addend && dst == a.y) { const float4 ad = amx_ld4(a.addend + o); v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w; }
                amx_st4(dst + o, v);
            }
        }
        if (m0 + MH < MTW) amx_wave_sync();                      // the region is rewritten by the next pair of rows
    }
    AMX_TICK(13);
}

template <int TAPS, int NT, int MAXHALO, bool EXACT, int MTW, int EPI, bool TAIL = false, int LAT = 0, int REM = 0>
static int launch_conv_fwd(const ConvFwdArgs& a, hipStream_t stream) {
    constexpr int NB = NT * 16 + REM * 4;
    const int halo = (TAPS == 9) ? (LAT ? 1 : a.dil) : 0;
    const int I = TILE + 2 * halo;
    const int plane = amx_round_up((4 * MTW + 2 * halo) * I, 16);
    size_t lds_w = (size_t)TAPS * KG * NB * 4 * sizeof(float);
    if (lds_w < (size_t)8 * NB * sizeof(float)) lds_w = (size_t)8 * NB * sizeof(float);
    size_t lds = ((size_t)KG * plane * 4 * sizeof(float) + lds_w) ;
    {   // the epilogue's transposition buffers: 4 waves x min(MTW, 2) rows x 16 pixels x NB couts
        const size_t epi = (size_t)4 * (MTW < 2 ? MTW : 2) * TILE * (NB + AMX_CONV_EPI_PAD) * sizeof(float);
        if (lds < epi) lds = epi;
    }
    // (REM: one cout block covers every stored channel)
    dim3 grid(a.tiles_x * a.tiles_y * a.N * (LAT ? (a.xpack ? LAT : LAT * LAT) : 1), REM ? 1 : amx_ceil_div(a.cop, NT * 16));
    AMX_ALLOW_160K_LDS(conv_fwd_kernel<TAPS, NT, MAXHALO, EXACT, MTW, EPI, TAIL, LAT, REM>);
    AMX_LAUNCH((conv_fwd_kernel<TAPS, NT, MAXHALO, EXACT, MTW, EPI, TAIL, LAT, REM>), grid, dim3(256), lds, stream, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

// wave-specialised kernel of the thin plain-3x3 layers (conv_ws.hip)
bool amx_conv_ws_supported(const ConvFwdArgs& a, int taps, int dil, int strip, float in_slope0, float in_slope1);
int amx_conv_launch_ws(ConvFwdArgs& a, hipStream_t s);
// dispatchers of the three instantiation units (nt in {1,2,4}; th in {8,16})
int amx_conv_launch_1x1(ConvFwdArgs& a, int nt, bool tail, hipStream_t s);
int amx_conv_launch_3x3(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s);
int amx_conv_launch_3x3_head(ConvFwdArgs& a, int nt, bool tail, hipStream_t s);
int amx_conv_launch_dil(ConvFwdArgs& a, int nt, bool tail, hipStream_t s);
int amx_conv_launch_lat2(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s);   // lattice mode, dilation 2 / 4 / 6
int amx_conv_launch_lat4(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s);
int amx_conv_launch_lat6(ConvFwdArgs& a, int nt, int th, bool tail, hipStream_t s);
// remainder-column classes (REM: 16 + 3 x 4 and 3 x 16 + 4 columns, 8-row tiles): conv_fwd_rem.hip, conv_fwd_lat{2,4,6}_rem.hip
int amx_conv_launch_3x3_rem(ConvFwdArgs& a, int nt, int rem, bool tail, hipStream_t s);
int amx_conv_launch_lat_rem(ConvFwdArgs& a, int dil, int nt, int rem, bool tail, bool dsum, hipStream_t s);
int amx_conv_launch_lat2_rem(ConvFwdArgs& a, int nt, int rem, bool tail, bool dsum, hipStream_t s);
int amx_conv_launch_lat4_rem(ConvFwdArgs& a, int nt, int rem, bool tail, bool dsum, hipStream_t s);
int amx_conv_launch_lat6_rem(ConvFwdArgs& a, int nt, int rem, bool tail, bool dsum, hipStream_t s);
int amx_conv_launch_lat2_dsum(ConvFwdArgs& a, bool tail, hipStream_t s);           // + fused DilatedBlock sum (eval)
int amx_conv_launch_lat4_dsum(ConvFwdArgs& a, bool tail, hipStream_t s);
int amx_conv_launch_lat6_dsum(ConvFwdArgs& a, bool tail, hipStream_t s);
