// wgrad.hip — C ABI and launch plan of the weight-gradient kernels (kernel: wgrad_kernel.h; lattice-mode
// instantiations for dilations 2 / 4 / 6: wgrad_lat{2,4,6}.hip).
#include "wgrad_kernel.h"

#ifdef AMX_WGRAD_PROFILE
void* amx_wgrad_profile_buffer = nullptr;
extern "C" int amx_wgrad_set_profile_buffer(void* buf) { amx_wgrad_profile_buffer = buf; return 0; }
#endif

struct WgradPlan { int NT, WM, WN, WK, ksplit, rows, ci_pad, co_pad, th, lat; };

static WgradPlan plan_wgrad(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    WgradPlan pl;
    const int odil = dil;
    pl.ci_pad = amx_round_up(Cin_s, 16);
    pl.co_pad = amx_round_up(cout, 16);
    pl.NT = pl.co_pad >= 32 ? 2 : 1;
    pl.WM = pl.ci_pad >= 64 ? 4 : (pl.ci_pad >= 32 ? 2 : 1);
    const bool lat = amx_lattice_mode(taps, dil);           // dilations 2 / 4 / 6 run the plain-3x3 plan on sub-lattices
    if (lat) dil = 1;
    if (taps == 9 && dil > 1) pl.WM = 1;
    const int rem = 4 / pl.WM;
    const int co_tiles = pl.co_pad / (16 * pl.NT);                // wave-level co tiles needed
    pl.WN = 1;
    while (pl.WN * 2 <= rem && pl.WN * 2 <= co_tiles && 16 * pl.NT * pl.WN * 2 <= 64) pl.WN *= 2;
    pl.WK = rem / pl.WN;
    const int blocks = amx_ceil_div(pl.ci_pad, 16 * pl.WM) * amx_ceil_div(pl.co_pad, 16 * pl.NT * pl.WN);
    // Tile height / split-K target.  Stand-alone, layers with <= 32 input channels run 15-25 % faster with 4-row
    // tiles and two workgroups per CU (profiles/r01_wgrad_sweep.md), but inside the training step the weight
    // gradients run on the side stream next to the data-gradient convolutions, and there the lighter
    // one-workgroup-per-CU launch wins (20.79 vs 21.43 ms/step, interleaved A/B with AMX_WGRAD_LIGHT): the step
    // is what is optimised.  AMX_WGRAD_LIGHT=<wgs> switches the stand-alone optimum on for experiments.
    const AmxKnobs& kn = amx_knobs();
    pl.th = (taps == 9 && dil == 1 && pl.NT == 2) ? 4 : 8;
    { const int v = kn.wgrad_th; if (v == 8 || (v == 4 && taps == 9 && dil == 1)) pl.th = v; }      // AMX_WGRAD_TH (tests, A/B)
    // 16 -> 32 channels (U-Net c2.0) on the wave-specialised kernel: with 4-row tiles a consumer wave has ONE row (72 MFMAs,
    // ~1 us) per tile and the producers' loads of tile k+2 are not back when tile k+1 must be staged; 8-row tiles double the
    // prefetch distance and cut the halo re-read from 1.69x to 1.41x (profiles/r04_wgrad_ws.md)
    if (taps == 9 && dil == 1 && !lat && pl.NT == 2 && pl.WM == 1 && pl.WN == 1 && amx_wgrad_ws_mask() & 1 && !kn.wgrad_th)
        pl.th = 8;
    if (lat) pl.th = pl.NT == 2 ? 4 : 8;                    // the instantiated lattice classes
    if (pl.WK > pl.th) pl.WK = pl.th;
    pl.lat = lat ? odil : 0;
    const int ls = lat ? odil : 1;
    const int ntiles = amx_ceil_div(amx_ceil_div(W, ls), TW) * amx_ceil_div(amx_ceil_div(H, ls), pl.th) * N * ls * ls;
    // split-K workgroups: one per CU (256: 20.98 ms/step, 512: 21.5, 1024: 21.6, 128: 25.2)
    int target = 256;
    if (kn.wgrad_wgs >= 1) target = kn.wgrad_wgs;            // AMX_WGRAD_WGS (tests: few workgroups, many tiles each)
    int ks = amx_ceil_div(target, blocks);
    if (ks > ntiles) ks = ntiles;
    if (ks < 1) ks = 1;
    pl.ksplit = ks;
    pl.rows = ks * pl.WK;
    return pl;
}

extern "C" int amx_conv2d_wgrad_ksplit(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    return plan_wgrad(N, H, W, Cin_s, cout, taps, dil).ksplit;
}

// rows / floats of the partial buffer amx_conv2d_wgrad needs
extern "C" int amx_conv2d_wgrad_rows(int N, int H, int W, int Cin_s, int cout, int taps, int dil) {
    return plan_wgrad(N, H, W, Cin_s, cout, taps, dil).rows;
}

static int wgrad_common(const float* x0, const float* sc0, const float* sh0, int C0s,
                        const float* x1, const float* sc1, const float* sh1, int C1s,
                        const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                        int taps, int dil, const float* aux, const float* k1, const float* k2, const float* k3,
                        float bslope, float* bpart, void* stream, float in_slope0 = 1.f, float in_slope1 = 1.f);

extern "C" int amx_conv2d_wgrad(const float* x0, const float* sc0, const float* sh0, int C0s,
                                const float* x1, const float* sc1, const float* sh1, int C1s,
                                const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                                int taps, int dil, void* stream) {
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dpre, Dos, part, N, H, W, cout, taps, dil, nullptr,
                        nullptr, nullptr, nullptr, 1.f, nullptr, stream);
}

// Weight gradient with the layer's BatchNorm/LeakyReLU backward fused into the dy loader and the bias gradient's
// partial sums (bpart: [amx_conv2d_wgrad_ksplit][round_up(cout,16)]) produced on the way.
extern "C" int amx_conv2d_wgrad_fused(const float* x0, const float* sc0, const float* sh0, int C0s,
                                      const float* x1, const float* sc1, const float* sh1, int C1s,
                                      const float* dy, const float* aux, const float* k1, const float* k2,
                                      const float* k3, float bslope, int Dos, float* part, float* bpart,
                                      int N, int H, int W, int cout, int taps, int dil, void* stream) {
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dy, Dos, part, N, H, W, cout, taps, dil, aux, k1, k2,
                        k3, bslope, bpart, stream);
}

// amx_conv2d_wgrad_fused for a layer whose input is read as LeakyReLU(affine(x)) (ResBlock's second conv)
extern "C" int amx_conv2d_wgrad_act(const float* x0, const float* sc0, const float* sh0, float in_slope0, int C0s,
                                    const float* x1, const float* sc1, const float* sh1, float in_slope1, int C1s,
                                    const float* dy, const float* aux, const float* k1, const float* k2,
                                    const float* k3, float bslope, int Dos, float* part, float* bpart,
                                    int N, int H, int W, int cout, int taps, int dil, void* stream) {
    if (!(in_slope0 > 0.f) || !(in_slope1 > 0.f)) AMX_BADARG(8);
    return wgrad_common(x0, sc0, sh0, C0s, x1, sc1, sh1, C1s, dy, Dos, part, N, H, W, cout, taps, dil, aux, k1, k2,
                        k3, bslope, bpart, stream, in_slope0, in_slope1);
}

extern "C" int amx_conv2d_wgrad_ksplit(int N, int H, int W, int Cin_s, int cout, int taps, int dil);

static int wgrad_common(const float* x0, const float* sc0, const float* sh0, int C0s,
                        const float* x1, const float* sc1, const float* sh1, int C1s,
                        const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                        int taps, int dil, const float* aux, const float* k1, const float* k2, const float* k3,
                        float bslope, float* bpart, void* stream, float in_slope0, float in_slope1) {
    if (!x0 || !dpre || !part) AMX_BADARG(1);
    if ((k1 == nullptr) != (k2 == nullptr) || (k1 == nullptr) != (k3 == nullptr)) AMX_BADARG(7);
    if (N <= 0 || H <= 0 || W <= 0 || cout <= 0) AMX_BADARG(2);
    if ((C0s & 3) || (C1s & 3) || (Dos & 3) || C0s <= 0 || Dos < cout) AMX_BADARG(3);
    if (taps != 1 && taps != 9) AMX_BADARG(4);
    if (taps == 9 && (dil < 1 || dil > 6)) AMX_BADARG(5);
    if ((x1 == nullptr) != (C1s == 0)) AMX_BADARG(6);
    const WgradPlan pl = plan_wgrad(N, H, W, C0s + C1s, cout, taps, dil);
    WgradArgs a;
    a.x0 = x0; a.sc0 = sc0; a.sh0 = sh0; a.C0s = C0s;
    a.x1 = x1; a.sc1 = sc1; a.sh1 = sh1; a.C1s = C1s;
    a.in_slope0 = in_slope0; a.in_slope1 = in_slope1;
    a.dpre = dpre; a.Dos = Dos; a.part = part;
    a.prof = nullptr;
#ifdef AMX_WGRAD_PROFILE
    a.prof = (unsigned long long*)amx_wgrad_profile_buffer;
#endif
    a.aux = aux; a.k1 = k1; a.k2 = k2; a.k3 = k3; a.bslope = bslope; a.bpart = bpart;
    a.N = N; a.H = H; a.W = W; a.dil = dil;
    a.ci_pad = pl.ci_pad; a.co_pad = pl.co_pad;
    a.WN = pl.WN; a.WK = pl.WK; a.ksplit = pl.ksplit;
    a.tiles_x = amx_ceil_div(W, TW); a.tiles_y = amx_ceil_div(H, pl.th);
    if (pl.lat) { a.tiles_x = amx_ceil_div(amx_ceil_div(W, pl.lat), TW); a.tiles_y = amx_ceil_div(amx_ceil_div(H, pl.lat), pl.th); a.dil = 1; }
    a.co_blocks = amx_ceil_div(pl.co_pad, 16 * pl.NT * pl.WN);
    hipStream_t s = (hipStream_t)stream;
#define WG_DISPATCH(T, H_, TH_)                                                   \
    if (pl.NT == 1) {                                                              \
        if (pl.WM == 1) return launch_wgrad<T, 1, 1, H_, TH_>(a, s);                \
        if (pl.WM == 2) return launch_wgrad<T, 1, 2, H_, TH_>(a, s);                \
        return launch_wgrad<T, 1, 4, H_, TH_>(a, s);                                \
    } else {                                                                       \
        if (pl.WM == 1) return launch_wgrad<T, 2, 1, H_, TH_>(a, s);                \
        if (pl.WM == 2) return launch_wgrad<T, 2, 2, H_, TH_>(a, s);                \
        return launch_wgrad<T, 2, 4, H_, TH_>(a, s);                                \
    }
    if (amx_wgrad_ws_supported(a, taps, dil, pl.lat, pl.NT, pl.WM, pl.th)) return amx_wgrad_launch_ws(a, pl.NT, pl.WM, pl.th, s);
    if (pl.lat == 2) return amx_wgrad_launch_lat2(a, pl.NT, pl.WM, s);
    if (pl.lat == 4) return amx_wgrad_launch_lat4(a, pl.NT, pl.WM, s);
    if (pl.lat == 6) return amx_wgrad_launch_lat6(a, pl.NT, pl.WM, s);
    if (taps == 1) { WG_DISPATCH(1, 0, 8) }
    if (dil == 1) {
        if (pl.th == 4) { WG_DISPATCH(9, 1, 4) }
        WG_DISPATCH(9, 1, 8)
    }
    if (pl.NT == 1) return launch_wgrad<9, 1, 1, 6, 8>(a, s);
    return launch_wgrad<9, 2, 1, 6, 8>(a, s);
#undef WG_DISPATCH
}
