// conv_fwd.hip — C ABI and launch plan of the MFMA convolution (kernel: conv_kernel.h; instantiations:
// conv_fwd_1x1.hip, conv_fwd_3x3.hip, conv_fwd_dil.hip, conv_fwd_lat{2,4,6}.hip).
#include "conv_kernel.h"

#ifdef AMX_CONV_PROFILE
static void* amx_conv_profile_buffer = nullptr;
extern "C" int amx_conv_set_profile_buffer(void* buf) { amx_conv_profile_buffer = buf; return 0; }
#endif

struct ConvPlan { int nt, th, rem; };
extern "C" int amx_conv2d_dgrad_fused_supported(int Cs, int Y0s, int Y1s, int N, int H, int W, int taps, int dil);

// Tile plan, from the per-shape measurements in profiles/r01_conv_variants.md: this kernel is fastest with MANY
// small co-resident workgroups (they hide each other's prologue / staging / epilogue), so the default tile is
// 8 rows x 16 pixels x 32 couts; only deep-K wide layers (Cin >= 128, Cout >= 64) pay for 16 rows x 64 couts.
// Rejected by measurement: 32-row tiles (-25 %), 4-row tiles (-5 %), a two-stage LDS pipeline (-5..25 %),
// occupancy forced through __launch_bounds__ (spills, -20..60 %).
// Dilations 2 / 4 / 6 (every dilation the reference's nets use, fcnn.py:186-200) run in lattice mode: d*d plain 3x3
// convolutions on the residue-class sub-images (conv_kernel.h).  AMX_CONV_LATTICE=0 selects the round-1 halo-class
// kernels of conv_fwd_dil.hip instead (in-process A/B; dilations 3 and 5 always use those).
bool amx_lattice_mode(int taps, int dil) {
    if (taps != 9 || (dil != 2 && dil != 4 && dil != 6)) return false;
    return amx_knobs().conv_lattice != 0;
}

// Stride of the x-packed lattice tiles for an image width and dilation (conv_kernel.h, ConvFwdArgs::xpack), 0 when the
// launch keeps one sub-image per tile axis: packing must save tile columns, and launches that write batch statistics
// keep the residue-class row order the merge kernels expect.
static int lattice_xpack(int W, int dil, bool has_stats) {
    if (!amx_knobs().conv_xpack || has_stats || !amx_lattice_mode(9, dil)) return 0;
    const int P = amx_ceil_div(W, dil) + 1, packed = amx_ceil_div(dil * P - 1, TILE);
    return (packed < dil * amx_ceil_div(amx_ceil_div(W, dil), TILE) && (long)dil * P < 65536L) ? P : 0;
}
extern "C" int amx_conv2d_lattice_xpack(int W, int dil, int has_stats) { return lattice_xpack(W, dil, has_stats != 0); }

static ConvPlan plan_conv(int Cin_s, int cout, int taps, int dil, int H, bool allow_rem = true) {
    ConvPlan pl;
    pl.rem = 0;
    const int cop = amx_round_up(cout, 16);
    const int nchunk = amx_ceil_div(Cin_s, 4 * KG);
    const bool small = taps == 9 && (dil == 1 || amx_lattice_mode(taps, dil));   // 8-row tiles exist for the plain geometry
    if (cop <= 16) { pl.nt = 1; pl.th = 16; }
    else if (cop >= 64 && (nchunk >= 8 || !small)) { pl.nt = 4; pl.th = 16; }
    else { pl.nt = 2; pl.th = small ? 8 : 16; }
    const AmxKnobs& kn = amx_knobs();
    { const int v = kn.conv_nt; if ((v == 1 || v == 2 || v == 4) && v * 16 <= cop) pl.nt = v; }       // AMX_CONV_NT (tests, A/B)
    { const int v = kn.conv_th; if (v == 16 || (v == 8 && small)) pl.th = v; }                        // AMX_CONV_TH
    // Round 4: widths of 25 / 50 filters (dilnet; 28 / 52 stored channels) run as 16 + 3 x 4 / 3 x 16 + 4 columns in ONE cout
    // block — the 4-wide remainder blocks on v_mfma_f32_4x4x1 (conv_kernel.h, REM) — instead of 32 / 2 x 32 padded
    // columns.  AMX_CONV_REM=0 switches it off (A/B); the fused classification head keeps the power-of-two plan.
    if (allow_rem && small && pl.th == 8 && pl.nt == 2) {
        const int c4 = amx_round_up(cout, 4);
        const bool on = kn.conv_rem != 0;
        if (on && c4 == 52) { pl.nt = 3; pl.rem = 1; }
        if (on && c4 == 28) { pl.nt = 1; pl.rem = 3; }
    }
    return pl;
}

// The classification head can be fused into a plain 3x3 layer whose plan is one of the two thin classes with ONE cout
// block: 16 couts x 16-row tiles (U-Net's last layer) or <= 32 couts x 8-row tiles (dilnet's).
static bool head_supported(int Cin_s, int cout, int taps, int dil, int H) {
    if (taps != 9 || dil != 1) return false;
    // (the fused head keeps the power-of-two column plan: the 28-column class with the head — 16 + 3 x 4 columns, three
    //  waves per SIMD — measured SLOWER than 32 padded columns at four, 1.152 vs 1.134 ms per dilnet frame,
    //  profiles/r04_logs/r04_dilnet_rem_head_ab.log)
    const ConvPlan pl = plan_conv(Cin_s, cout, taps, dil, H, false);
    const int cop = amx_round_up(cout, 16);
    return (pl.nt == 1 && pl.th == 16 && cop == 16) || (pl.nt == 2 && pl.th == 8 && cop == 32);
}
extern "C" int amx_conv2d_head_supported(int Cin_s, int cout, int taps, int dil, int H) {
    return head_supported(Cin_s, cout, taps, dil, H) ? 1 : 0;
}

// The sum of a DilatedBlock can be fused into its last layer when that layer runs in lattice mode on the 8-row,
// 32-couts-per-block class (dilnet's 25 / 50-filter blocks).
static bool dsum_supported(int Cin_s, int cout, int taps, int dil, int H) {
    if (!amx_lattice_mode(taps, dil)) return false;
    const ConvPlan pl = plan_conv(Cin_s, cout, taps, dil, H);
    return (pl.nt == 2 && pl.th == 8) || (pl.rem && pl.th == 8);
}
extern "C" int amx_conv2d_dsum_supported(int Cin_s, int cout, int taps, int dil, int H) {
    return dsum_supported(Cin_s, cout, taps, dil, H) ? 1 : 0;
}

// epilogue extras of the eval-mode fusions (classification head / DilatedBlock sum)
struct ConvEpi {
    const float* hw = nullptr; const float* hb = nullptr; float* hout = nullptr; int hK = 0, hmode = 0;
    const float* ds_a[3] = {nullptr, nullptr, nullptr};
    const float* ds_sc[4] = {nullptr, nullptr, nullptr, nullptr}; const float* ds_sh[4] = {nullptr, nullptr, nullptr, nullptr};
    int nds = 0; float ds_inv_slope = 1.f;
};

// loader extras of the data gradient that forms dpre from (dy, a) on the fly (conv_kernel.h: ConvFwdArgs::bw_*)
struct ConvBwdLoad { const float* aux; const float* k1; const float* k2; const float* k3; float slope;
                     const float* bs_a = nullptr; float* bs_part = nullptr; };
int amx_conv_ws_bsum_rows(int N, int H, int W, int cout);      // conv_ws.hip

// C ABI — see include/atomai_amd.h for the contract.
static int conv2d_common(const float* x0, const float* sc0, const float* sh0, int C0s,
                         const float* x1, const float* sc1, const float* sh1, int C1s,
                         const float* wpk, const float* bias, const float* addend,
                         float* y, int Y0s, float* y1, int Y1s, float* stats,
                         int N, int H, int W, int cout, int taps, int dil, float slope, void* stream,
                         float in_slope0 = 1.f, float in_slope1 = 1.f, const ConvEpi* epi = nullptr,
                         const ConvBwdLoad* bw = nullptr) {
    const float* hw = epi ? epi->hw : nullptr; const float* hb = epi ? epi->hb : nullptr;
    float* hout = epi ? epi->hout : nullptr; const int hK = epi ? epi->hK : 0, hmode = epi ? epi->hmode : 0;
    const int nds = epi ? epi->nds : 0;
    if (!x0 || !wpk || (!y && !hout)) AMX_BADARG(1);
    if (N <= 0 || H <= 0 || W <= 0 || cout <= 0) AMX_BADARG(2);
    if ((C0s & 3) || (C1s & 3) || (Y0s & 3) || (Y1s & 3) || C0s <= 0) AMX_BADARG(3);
    if (taps != 1 && taps != 9) AMX_BADARG(4);
    if (taps == 9 && (dil < 1 || dil > 6)) AMX_BADARG(5);
    if ((x1 == nullptr) != (C1s == 0)) AMX_BADARG(6);
    if ((y1 == nullptr) != (Y1s == 0)) AMX_BADARG(7);
    if ((long)N * H * W >= 2147483647L) AMX_BADARG(2);           // pixel offsets are 32-bit
    if (stats && addend) AMX_BADARG(10);                          // statistics describe the un-accumulated output
    ConvFwdArgs a;
    a.x0 = x0; a.sc0 = sc0; a.sh0 = sh0; a.C0s = C0s;
    a.x1 = x1; a.sc1 = sc1; a.sh1 = sh1; a.C1s = C1s;
    a.in_slope0 = in_slope0; a.in_slope1 = in_slope1;
    a.wpk = wpk; a.bias = bias; a.addend = addend;
    a.y = y; a.Y0s = Y0s; a.y1 = y1; a.Y1s = Y1s; a.stats = stats;
    a.hw = hw; a.hb = hb; a.hout = hout; a.hK = hK; a.hmode = hmode;     // fused classification head (eval) or nullptr
    a.nds = nds; a.ds_inv_slope = epi ? epi->ds_inv_slope : 1.f;         // fused DilatedBlock sum (eval) or 0
    for (int l = 0; l < 3; ++l) a.ds_a[l] = (epi && l < nds) ? epi->ds_a[l] : nullptr;
    for (int l = 0; l < 4; ++l) { a.ds_sc[l] = (epi && l <= nds) ? epi->ds_sc[l] : nullptr; a.ds_sh[l] = (epi && l <= nds) ? epi->ds_sh[l] : nullptr; }
    a.bw_aux = bw ? bw->aux : nullptr; a.bw_k1 = bw ? bw->k1 : nullptr; a.bw_k2 = bw ? bw->k2 : nullptr;
    a.bw_k3 = bw ? bw->k3 : nullptr; a.bw_slope = bw ? bw->slope : 1.f;
    a.bs_a = bw ? bw->bs_a : nullptr; a.bs_part = bw ? bw->bs_part : nullptr;
    if ((a.bs_a == nullptr) != (a.bs_part == nullptr) || (a.bs_a && (y1 || addend))) AMX_BADARG(17);
    a.prof = nullptr;
#ifdef AMX_CONV_PROFILE
    a.prof = (unsigned long long*)amx_conv_profile_buffer;               // dev build: per-wave phase timestamps
#endif
    // XCD-aware block order (conv_kernel.h): the cout blocks of a tile — and, in lattice mode, the d*d residue classes of
    // a tile position, which share cache lines — become neighbours on ONE XCD's L2 instead of being dealt round-robin
    // over the eight XCDs.  AMX_CONV_XCD: 0 off, 1 every launch (default since round 6), 2 launches with more than one cout
    // block, 3 dilated launches only.  The counters showed the dilated layers of dilnet fetching 3.3x (dilation 2 / 4) and 7.9x
    // (dilation 6) their input from HBM (profiles/r03_pmc_hbm_extra.md); with the XCD-aware order a dilnet frame goes
    // 1.282 -> 1.257 ms; plain 3x3 layers did not care in round 3 (U-Net step 17.92 vs 17.93 ms, profiles/r03_conv_xcd_ab.log); on
    // the round-6 step the order on EVERY launch is worth 0.1 ms (16.97 -> 16.86 ms, three interleaved repetitions all the same
    // sign, profiles/r06_logs/r06_xcd_default_ab.log) and cuts the conv family's HBM fetch by 22 % (r05_pmc_hbm_traffic.json).
    const int xm = amx_knobs().conv_xcd;
    a.xcd = xm == 1 || (xm == 2 && amx_round_up(cout, 16) > 32) || (xm == 3 && dil > 1);
    a.xpack = 0; a.xmagic = 0;
    a.N = N; a.H = H; a.W = W;
    a.cout = cout;
    a.cop = amx_round_up(cout, 16);
    a.nchunk = amx_ceil_div(C0s + C1s, 4 * KG);
    a.tail_kg = (C0s + C1s - (a.nchunk - 1) * 4 * KG) / 4;
    a.dil = dil; a.slope = slope;
    if (Y0s + Y1s < cout) AMX_BADARG(8);
    hipStream_t s = (hipStream_t)stream;
    const ConvPlan pl = plan_conv(C0s + C1s, cout, taps, dil, H, hout == nullptr);
    a.th = pl.th;
    a.tiles_x = amx_ceil_div(W, TILE); a.tiles_y = amx_ceil_div(H, pl.th);
    const bool tail = a.tail_kg < KG;                       // partial last chunk: cheaper tail path
    if (amx_conv_ws_supported(a, taps, dil, pl.th / 4, in_slope0, in_slope1)) return amx_conv_launch_ws(a, s);
    if (a.bw_aux) AMX_BADARG(14);                           // the on-load BatchNorm backward exists in conv_ws.hip only
    if (amx_lattice_mode(taps, dil)) {                          // tiles of the (largest) residue-class sub-image
        a.tiles_x = amx_ceil_div(amx_ceil_div(W, dil), TILE); a.tiles_y = amx_ceil_div(amx_ceil_div(H, dil), pl.th);
        if ((long)a.tiles_x * a.tiles_y * N * dil * dil >= 2147483647L) AMX_BADARG(2);
        a.dil = 1;
        // x-packed tiles (conv_kernel.h, ConvFwdArgs::xpack): whenever they are fewer than dil tile columns per sub-image
        // and no batch statistics are asked for (their rows are ordered by residue class) — every eval-mode launch and
        // every data gradient of a dilated layer whose sub-image width is not a multiple of 16 (dilation 6 at 512: 33 vs 36)
        if (const int P = lattice_xpack(W, dil, stats != nullptr)) {
            a.xpack = P; a.xmagic = (unsigned)((4294967296ULL + P - 1) / P); a.tiles_x = amx_ceil_div(dil * P - 1, TILE);
        }
        if (nds) {                                          // fused DilatedBlock sum: the 32-couts-per-block 8-row class only
            if (!dsum_supported(C0s + C1s, cout, taps, dil, H) || nds > 3 || stats || addend || y1 || hout) AMX_BADARG(13);
            for (int l = 0; l < nds; ++l) if (!a.ds_a[l]) AMX_BADARG(13);
            for (int l = 0; l <= nds; ++l) if (!a.ds_sc[l] || !a.ds_sh[l]) AMX_BADARG(13);
            if (pl.rem) return amx_conv_launch_lat_rem(a, dil, pl.nt, pl.rem, tail, true, s);
            if (dil == 2) return amx_conv_launch_lat2_dsum(a, tail, s);
            if (dil == 4) return amx_conv_launch_lat4_dsum(a, tail, s);
            return amx_conv_launch_lat6_dsum(a, tail, s);
        }
        if (pl.rem) return amx_conv_launch_lat_rem(a, dil, pl.nt, pl.rem, tail, false, s);
        if (dil == 2) return amx_conv_launch_lat2(a, pl.nt, pl.th, tail, s);
        if (dil == 4) return amx_conv_launch_lat4(a, pl.nt, pl.th, tail, s);
        return amx_conv_launch_lat6(a, pl.nt, pl.th, tail, s);
    }
    if (nds) AMX_BADARG(13);                                // (the fused sum exists for the lattice classes only)
    if (hout) {                                             // fused head: plain 3x3, one cout block, the two thin classes
        if (!head_supported(C0s + C1s, cout, taps, dil, H) || !hw || !hb || hK < 1 || hK > 3 || hmode < 0 || hmode > 1
            || stats || addend || y1) AMX_BADARG(12);
        return amx_conv_launch_3x3_head(a, pl.nt, tail, s);
    }
    if (taps == 1) return amx_conv_launch_1x1(a, pl.nt, tail, s);
    if (dil == 1 && pl.rem) return amx_conv_launch_3x3_rem(a, pl.nt, pl.rem, tail, s);
    if (dil == 1) return amx_conv_launch_3x3(a, pl.nt, pl.th, tail, s);
    return amx_conv_launch_dil(a, pl.nt, tail, s);
}

extern "C" int amx_conv2d_fwd(const float* x0, const float* sc0, const float* sh0, int C0s,
                              const float* x1, const float* sc1, const float* sh1, int C1s,
                              const float* wpk, const float* bias, const float* addend,
                              float* y, int Y0s, float* y1, int Y1s, float* stats,
                              int N, int H, int W, int cout, int taps, int dil, float slope,
                              void* stream) {
    return conv2d_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, wpk, bias, addend, y, Y0s, y1, Y1s, stats, N, H, W,
                         cout, taps, dil, slope, stream);
}

// amx_conv2d_fwd with a LeakyReLU applied after the on-load BatchNorm affine of each source (1.0f == none):
// the conv -> BatchNorm -> LeakyReLU order of ResBlock (atomai/nets/blocks.py:199-214).
extern "C" int amx_conv2d_fwd_act(const float* x0, const float* sc0, const float* sh0, float in_slope0, int C0s,
                                  const float* x1, const float* sc1, const float* sh1, float in_slope1, int C1s,
                                  const float* wpk, const float* bias, const float* addend,
                                  float* y, int Y0s, float* y1, int Y1s, float* stats,
                                  int N, int H, int W, int cout, int taps, int dil, float slope, void* stream) {
    if (!(in_slope0 > 0.f) || !(in_slope1 > 0.f)) AMX_BADARG(11);
    return conv2d_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, wpk, bias, addend, y, Y0s, y1, Y1s, stats, N, H, W,
                         cout, taps, dil, slope, stream, in_slope0, in_slope1);
}

// amx_conv2d_fwd in eval mode with the network's classification head fused into the epilogue: the layer's activation
// a = lrelu(conv + bias) is NOT stored; out = head(a) where hw / hb hold the final 1x1 convolution with the layer's own
// eval-mode BatchNorm affine folded in (hw[k][c] = Wpx[k][c] * scale[c], hb[k] = bpx[k] + sum_c Wpx[k][c] * shift[c]).
// mode 0: logits NCHW [N][K][H][W]; mode 1: probabilities NHWC [N][H][W][K] (sigmoid if K == 1, else softmax).
// Replaces conv -> (BatchNorm eval) -> px -> sigmoid/softmax -> permute of SegPredictor.forward_ for the last layer
// (atomai/nets/fcnn.py:139-142, 224-226; atomai/predictors/predictor.py:219-229).
extern "C" int amx_conv2d_fwd_head(const float* x0, const float* sc0, const float* sh0, int C0s,
                                   const float* x1, const float* sc1, const float* sh1, int C1s,
                                   const float* wpk, const float* bias, const float* hw, const float* hb, float* out,
                                   int K, int mode, int N, int H, int W, int cout, float slope, void* stream) {
    if (!out) AMX_BADARG(12);
    ConvEpi e; e.hw = hw; e.hb = hb; e.hout = out; e.hK = K; e.hmode = mode;
    return conv2d_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, wpk, bias, nullptr, nullptr, amx_round_up(cout, 4),
                         nullptr, 0, nullptr, N, H, W, cout, 9, 1, slope, stream, 1.f, 1.f, &e);
}

// amx_conv2d_fwd in eval mode for the LAST layer of a DilatedBlock with the block's sum fused into the epilogue
// (atomai/nets/blocks.py:321-329: the block returns the sum of EVERY sub-layer output = per layer the convolution, its
// LeakyReLU and its BatchNorm): y = sum over layers l of [pre_l + a_l + (a_l * sc_l + sh_l)], pre_l = inverse LeakyReLU
// of a_l; prev: the n earlier layers' activations (device pointers, same shape as y); sc / sh: n + 1 eval-mode BatchNorm
// affines (this layer's last; zero vectors when the block has no BatchNorm).  This layer's own activation is not stored.
extern "C" int amx_conv2d_fwd_dsum(const float* x0, const float* sc0, const float* sh0, int C0s, const float* wpk,
                                   const float* bias, const float* const* prev, const float* const* sc,
                                   const float* const* sh, int n, float* y, int N, int H, int W, int cout, int dil,
                                   float slope, void* stream) {
    if (!prev || !sc || !sh || n < 1 || n > 3 || slope == 0.f) AMX_BADARG(13);
    ConvEpi e; e.nds = n; e.ds_inv_slope = 1.0f / slope;
    for (int l = 0; l < n; ++l) e.ds_a[l] = prev[l];
    for (int l = 0; l <= n; ++l) { e.ds_sc[l] = sc[l]; e.ds_sh[l] = sh[l]; }
    return conv2d_common(x0, sc0, sh0, C0s, nullptr, nullptr, nullptr, 0, wpk, bias, nullptr, y, amx_round_up(cout, 4),
                         nullptr, 0, nullptr, N, H, W, cout, 9, dil, slope, stream, 1.f, 1.f, &e);
}

// Data gradient: forward convolution of dpre (Cs stored channels) with the flipped / transposed weight image
// (amx_pack_weights mode 1); the two halves of a concatenated input receive their gradients through y / y1.
// (Round 1 also carried a variant with the BatchNorm / LeakyReLU backward formed in the loader and the source layers'
// backward statistics in the epilogue; it measured slower than the separate HBM-bound passes — DESIGN.md §5 — and
// was removed together with its register cost.)
extern "C" int amx_conv2d_dgrad(const float* dpre, int Cs, const float* wpk, const float* addend,
                                float* y, int Y0s, float* y1, int Y1s, int N, int H, int W, int taps, int dil,
                                void* stream) {
    return conv2d_common(dpre, nullptr, nullptr, Cs, nullptr, nullptr, nullptr, 0, wpk, nullptr, addend, y, Y0s, y1,
                         Y1s, nullptr, N, H, W, Y0s + Y1s, taps, dil, 1.f, stream);
}

// Data gradient with the layer's BatchNorm / LeakyReLU backward formed by the loader (no materialised dpre): only for
// launches the wave-specialised kernel takes (amx_conv2d_dgrad_fused_supported), whose producer waves have the
// registers and issue slots for the second tensor.  Replaces amx_bn_bwd_apply + amx_conv2d_dgrad for those layers.
extern "C" int amx_conv2d_dgrad_fused(const float* dy, const float* aux, const float* k1, const float* k2,
                                      const float* k3, float bslope, int Cs, const float* wpk, float* y, int Y0s,
                                      float* y1, int Y1s, int N, int H, int W, int taps, int dil, void* stream) {
    if (!aux) AMX_BADARG(14);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(15);
    ConvBwdLoad bw{aux, k1, k2, k3, bslope};
    return conv2d_common(dy, nullptr, nullptr, Cs, nullptr, nullptr, nullptr, 0, wpk, nullptr, nullptr, y, Y0s, y1,
                         Y1s, nullptr, N, H, W, Y0s + Y1s, taps, dil, 1.f, stream, 1.f, 1.f, nullptr, &bw);
}
// amx_conv2d_dgrad_fused for a launch with ONE output that is the complete gradient of a conv -> LeakyReLU -> BatchNorm
// layer's output: bs_a = that layer's saved activation (shape of y); bs_part receives amx_conv2d_dgrad_bsum_rows rows of
// (sum dy, sum dy * a) per channel — the input of amx_bn_bwd_finalize (stride Y0s), replacing amx_bn_bwd_reduce.
extern "C" int amx_conv2d_dgrad_fused_bsum(const float* dy, const float* aux, const float* k1, const float* k2,
                                           const float* k3, float bslope, int Cs, const float* wpk, float* y, int Y0s,
                                           int N, int H, int W, int taps, int dil, const float* bs_a, float* bs_part,
                                           void* stream) {
    if (!aux || !bs_a || !bs_part) AMX_BADARG(14);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(15);
    ConvBwdLoad bw{aux, k1, k2, k3, bslope, bs_a, bs_part};
    return conv2d_common(dy, nullptr, nullptr, Cs, nullptr, nullptr, nullptr, 0, wpk, nullptr, nullptr, y, Y0s, nullptr,
                         0, nullptr, N, H, W, Y0s, taps, dil, 1.f, stream, 1.f, 1.f, nullptr, &bw);
}
// rows of bs_part, 0 when the launch is not one the wave-specialised kernel takes (or AMX_BWD_SUMS=0)
extern "C" int amx_conv2d_dgrad_bsum_rows(int Cs, int Y0s, int N, int H, int W, int taps, int dil) {
    if (!amx_knobs().bwd_sums || !amx_conv2d_dgrad_fused_supported(Cs, Y0s, 0, N, H, W, taps, dil)) return 0;
    return amx_conv_ws_bsum_rows(N, H, W, Y0s);
}

extern "C" int amx_conv2d_dgrad_fused_supported(int Cs, int Y0s, int Y1s, int N, int H, int W, int taps, int dil) {
    if (Cs <= 0 || Y0s <= 0 || (taps != 1 && taps != 9)) return 0;
    if (!amx_knobs().bwd_fuse) return 0;
    ConvFwdArgs a = {};
    a.C0s = Cs; a.Y0s = Y0s; a.Y1s = Y1s; a.cout = Y0s + Y1s; a.N = N; a.H = H; a.W = W;
    a.bw_aux = reinterpret_cast<const float*>(&a);          // (any non-null value: the query dereferences nothing)
    return amx_conv_ws_supported(a, taps, dil, plan_conv(Cs, Y0s + Y1s, taps, dil, H).th / 4, 1.f, 1.f) ? 1 : 0;
}

// Height (in image rows) of the 16-pixel-wide strip one partial-statistics row of amx_conv2d_fwd covers for this
// layer, and the number of such rows it writes: rows x 2 x round_up(cout,16).
extern "C" int amx_conv2d_tile_h(int Cin_s, int cout, int taps, int dil, int H) {
    return plan_conv(Cin_s, cout, taps, dil, H).th / 4;      // one statistics row per WAVE: a strip of th/4 image rows
}
extern "C" int amx_conv2d_num_tiles(int N, int H, int W, int th) {
    return amx_ceil_div(W, TILE) * amx_ceil_div(H, th) * N;
}
// Layout of the partial-statistics rows amx_conv2d_fwd writes for this layer: the dilation d if the rows are ordered
// [n][ry][rx][strip][tx] over the d*d residue-class sub-images (amx_bn_stats_merge mode 3), 0 for the plain
// [n][strip][tx] order (mode 0); and the number of rows.
extern "C" int amx_conv2d_stats_lattice(int taps, int dil) { return amx_lattice_mode(taps, dil) ? dil : 0; }
extern "C" int amx_conv2d_stats_rows(int Cin_s, int cout, int taps, int dil, int N, int H, int W) {
    const int th = plan_conv(Cin_s, cout, taps, dil, H).th / 4;
    if (!amx_lattice_mode(taps, dil)) return amx_ceil_div(W, TILE) * amx_ceil_div(H, th) * N;
    return amx_ceil_div(amx_ceil_div(W, dil), TILE) * amx_ceil_div(amx_ceil_div(H, dil), th) * dil * dil * N;
}
