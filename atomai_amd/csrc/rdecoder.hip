// rdecoder.hip — the rVAE "spatial decoder": an MLP evaluated at EVERY pixel coordinate.
//
//   rDecoderNet.forward + coord_latent.forward            atomai/nets/ed.py:626-642, 672-687
//     h0 = [tanh](Wc (x',y') + bc + Wz z)                 (tanh unless skip)
//     h_l = tanh(W_l h_{l-1} + b_l) [+ h0 if skip]        l = 1..NL
//     out = Wo h_NL + bo                                  one value per pixel (grayscale)
//
// This is where the rVAE step spends its time: B*H*W rows x HID hidden units
// (2.1 M x 128 at bs 512, 64x64 => 139 GFLOP forward), and the reference materialises every hidden
// activation in HBM (1 GB each).  Here one workgroup owns one sample and streams its pixels in tiles of
// MT; all hidden activations live ONLY in LDS:
//   * activation images in LDS are [feature/4][pixel][4] so that an MFMA operand fragment is one
//     conflict-free ds_read_b128 (same trick as conv_fwd.hip);
//   * the layer GEMM is evaluated transposed, D'[feature][pixel] = W · h^T, with A = W fragments held in
//     registers (one wave owns 16 output features) and B = activations from LDS, so the accumulator
//     fragment of a lane is 4 consecutive FEATURES of one pixel = exactly one b128 LDS store of the next
//     layer's operand image;
//   * backward recomputes the forward per tile, then runs output-layer, wgrad (MFMA, contraction over
//     pixels) and dgrad (MFMA with W^T) entirely from LDS; weight-gradient accumulators stay in registers
//     across all tiles of the sample and are written once as a per-sample partial row (summed by
//     amx_reduce_rows in fp64 -> deterministic).
#include "amx_device.h"
#include <cstdlib>

struct RDecArgs {
    const float* coords;   // theta == nullptr: [B][n][2] transformed pixel coordinates;  else the grid [n][2]
    const float* theta;    // [B][3] (phi, dx, dy) or nullptr: x' = x cos(phi) - y sin(phi) + dx, y' = x sin(phi) + y cos(phi) + dy
                           // formed per pixel IN the kernel (atomai/utils/coords.py:57-83 materialises (B, n, 2))
    float* dtheta;         // [B][3] gradient w.r.t. (phi, dx, dy)   (backward, theta mode)
    int C;                 // output channels (xrec / dxrec are [B][n][C], Wo is [C][HID], bo is [C])
    const float* z;        // [B][L]
    const float* Wc;       // [HID][2]
    const float* bc;       // [HID]
    const float* Wz;       // [HID][L]
    const float* W;        // [NL][HID][HID]
    const float* Wt;       // [NL][HID][HID] transposed copies (backward only)
    const float* b;        // [NL][HID]
    const float* Wo;       // [C][HID]
    const float* bo;       // [C]
    float* xrec;           // [B][n][C]                  (forward)
    const float* dxrec;    // [B][n][C]                  (backward)
    float* dcoords;        // [B][n][2]                  (backward, explicit-coordinate mode)
    float* dz;             // [B][L]
    float* pW;             // [B][NL][HID][HID] partial rows
    float* pb;             // [B][NL][HID]
    float* pWo;            // [B][C][HID]
    float* pbo;            // [B][C]
    float* pWc;            // [B][HID][2]
    float* pbc;            // [B][HID]
    float* pWz;            // [B][HID][L]
    int B, n, L, NL, skip;
    float* hsave;          // forward: [B][planes][HID/4][npad][4] activation images to keep for backward (planes = h0 [RD_SAVE_H0],
                           // h_1 .. h_NL), or nullptr
    const float* hsaved;   // backward: the same buffer (then the hidden layers are NOT recomputed), or nullptr
    int npad;              // pixels per plane of hsave (n rounded up to a multiple of 128)
    unsigned long long* prof;   // AMX_RDEC_PROFILE builds: per-wave phase clocks [workgroup][wave][8], or nullptr
};

// AMX_RDEC_PROFILE (dev builds only, tools/gpu_rdec_phases.py): every wave of the backward kernel accumulates the shader
// clocks it spends per phase of its tile loop: coordinate layer, hidden layers forward, output-layer backward, wgrad
// MFMAs, dgrad MFMAs, elementwise after dgrad, coordinate-layer backward, lifetime.
#ifdef AMX_RDEC_PROFILE
static void* amx_rdec_profile_buffer = nullptr;
extern "C" int amx_rdec_set_profile_buffer(void* buf) { amx_rdec_profile_buffer = buf; return 0; }
#define RD_TICK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt[i] += n_ - pl; pl = n_; } while (0)
#else
#define RD_TICK(i) do { } while (0)
#endif
#ifndef RD_PLANE_PAD
#define RD_PLANE_PAD 6        // slots (16 B) of padding per activation plane; 0 = the unpadded round-1 layout.  Round 5 sweep of the
                              // config-4 step (profiles/r05_logs/r05_rvae_pad_sweep.log, bit-identical gradients): pad 4 (rounds
                              // 2-4) 4.941 ms, 1 / 2 / 9 4.90, 5 4.877, 7 4.867, 3 4.858, 6 4.846, 8 / 12 5.13 (plane stride back
                              // on a multiple of 32 banks).  Also measured there: the weight gradient's B operands by
                              // ds_read_b128 with permuted accumulator columns (2 reads per k-step instead of 8 b32 reads,
                              // bit-identical): 4.961 vs 4.963 ms at pad 4, 4.906 vs 4.888 at pad 5 — the phase is not bound by
                              // its LDS instruction count; not kept.
#endif
#ifndef RD_SAVE_H0
#define RD_SAVE_H0 0          // experiment switch: 1 = the saved-activation mode also keeps h0 (the coordinate layer's output) so
                              // that the backward skips its 16 tanh + 32 FMA per lane and tile.  Measured SLOWER in the step
                              // (5.553 -> 5.620 ms, profiles/r03_rvae_h0_ab.log): 16 more prefetch registers (spill 120 ->
                              // 184 B in the <128, 64, 2> class) and 1.07 GB more HBM traffic each way.
#endif
#ifndef RD_COORD_PREFETCH
#define RD_COORD_PREFETCH 1   // the next tile's raw coordinate pair is fetched at the start of the current tile (0 = at its own start)
#endif
#ifndef RD_BWD_C1
#define RD_BWD_C1 1           // one-channel instantiation of the backward kernel (0 = MAXC accumulators for every patch)
#endif
#ifndef RD_COORD_PREFETCH_BWD
#define RD_COORD_PREFETCH_BWD 1
#endif
#ifndef RD_FWD_WPIPE
#define RD_FWD_WPIPE 1        // forward kernel: the next layer's weight fragments are fetched behind the current layer's MFMAs (0 = at its start)
#endif
#ifndef RD_FWD_ASYM
#define RD_FWD_ASYM 0         // experiment switch (round 6): asymmetric pair of co-resident forward workgroups, see the kernel
#endif
#define RD_PLANES(NL_) ((NL_) + (RD_SAVE_H0 ? 1 : 0))
#define MAXL 32           // latent dimensions (content latents + one-hot classes) the kernels keep in LDS
#define MAXC 4            // output channels the kernels are written for (grey-scale and RGB(A) patches)

// tanh(x) = 1 - 2 / (e^{2x} + 1) on the hardware exp / reciprocal units: 5 VALU instructions instead of the ~35 of
// the library tanhf, which otherwise costs as many issue cycles per tile as the layer's MFMAs (48 tanh per lane and
// tile).  Absolute error <= 2e-7 over the whole range (saturates correctly to +-1); set AMX_RDEC_EXACT_TANH to
// compile the library version instead.
// The bound is ABSOLUTE: near 0 the form cancels (1 - 2 / (t + 1) with t ~ 1 + 2x), so the relative error of tanh(1e-4) is
// ~1e-3; the activations are judged (and used: the derivative is 1 - h^2) on the scale of 1.  Tested element-wise through
// amx_rdec_tanh_probe over [-20, 20] including subnormal-adjacent inputs (tests/test_vae_gpu.py).
// AMX_EMU builds (CPU test tier) run tanhf by default; amx_emu_set_fast_tanh(1) switches them to the SAME algebraic form
// on expf / division, so the CPU tier exercises the cancellation structure the device runs (tests/test_vae_emulated.py).
#ifdef AMX_EMU
static int amx_emu_fast_tanh = 0;
extern "C" int amx_emu_set_fast_tanh(int on) { amx_emu_fast_tanh = on; return 0; }
#endif
static __device__ __forceinline__ float rd_tanh(float x) {
#if defined(AMX_EMU)
    if (amx_emu_fast_tanh) {
        const float t = expf(2.f * x);
        return 1.f - 2.f / (t + 1.f);
    }
    return tanhf(x);
#elif defined(AMX_RDEC_EXACT_TANH)
    return tanhf(x);
#else
    const float t = __expf(2.f * x);
    return 1.f - __fdividef(2.f, t + 1.f);
#endif
}

// the decoder's activation function, element-wise (tests: the absolute error bound claimed above)
__global__ __launch_bounds__(256) void rdec_tanh_probe_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = rd_tanh(x[i]);
}
extern "C" int amx_rdec_tanh_probe(const float* x, float* y, long n, void* stream) {
    if (!x || !y || n <= 0) AMX_BADARG(1);
    AMX_LAUNCH(rdec_tanh_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    AMX_CHECK_LAUNCH();
    return 0;
}

// --------------------------------------------------------------------------------------------------
// shared pieces
template <int HID, int MT>
struct Geo {
    static constexpr int NW = HID / 16;          // waves
    static constexpr int NT = 64 * NW;           // threads
    static constexpr int KG = HID / 4;           // feature groups of 4
    static constexpr int PT = MT / 16;           // pixel tiles per wave GEMM
    // Plane stride of an activation image [kg][pixel][4] in 16-byte slots: MT + RD_PLANE_PAD.  With a stride of MT (a
    // multiple of 32 banks) the scalar operand reads of the in-kernel weight gradient — lane (p, g) reads feature 16w+p of
    // pixel 4s+g, i.e. planes (16w+p)>>2 — hit each bank four times; padding spreads the planes over the banks (the sweep
    // behind the value: RD_PLANE_PAD above).  The float4 fragment reads / writes of the other phases stay conflict free (8 consecutive lanes
    // still cover 128 contiguous bytes).  tools/gpu_rdec_phases.py: the wgrad phase was LDS-bound (23.2 k clocks per
    // tile for 16.4 k of MFMA time).
    static constexpr int PS = MT + RD_PLANE_PAD;
    static constexpr int BUF = HID * PS;         // floats per activation image (KG planes of PS slots)
    static constexpr int SPT = KG * MT / NT;     // (kg,pixel) slots per thread  (= MT/16)
    static constexpr int TPP = NT / MT;          // threads per pixel in the output stage
};

// Raw coordinate pair of this thread's pixel of the tile at pix0 (theta mode: the shared grid point, else the sample's own
// transformed point); clamped to a valid pixel past the end.  Round 5: the kernels fetch the NEXT tile's pair at the start
// of the current tile (RD_COORD_PREFETCH) — the coordinate layer is the first thing a tile does, and its two dependent
// global loads were an exposed L2 round trip per tile.
struct RawXY { float x, y; };
template <int MT>
__device__ __forceinline__ RawXY load_raw_xy(const RDecArgs& a, int bidx, int pix0, int tid, bool theta_mode) {
    int q = pix0 + tid % MT;
    q = q < a.n ? q : a.n - 1;
    const float* c = theta_mode ? a.coords + (size_t)q * 2 : a.coords + ((size_t)bidx * a.n + q) * 2;
    RawXY r; r.x = c[0]; r.y = c[1];
    return r;
}

// h0 tile -> dst (KG layout).  Also stores x',y' per pixel when s_xy != nullptr.
template <int HID, int MT>
__device__ __forceinline__ void coord_layer(const RDecArgs& a, int bidx, int pix0, const float* s_zc,
                                            float* dst, float* s_xy, const float* s_th, int tid,
                                            float* gsave = nullptr, const float* wc = nullptr, const RawXY* pre = nullptr) {
    if (!wc) wc = a.Wc;                          // (the kernels pass their LDS copy: s_wc)
    using G = Geo<HID, MT>;
    static_assert(G::NT % MT == 0, "a thread's slots share one pixel");
    // every slot of a thread is the same pixel (NT is a multiple of MT): its transformed coordinates once per tile
    const int p = tid % MT, q = pix0 + p;
    float xx = 0.f, yy = 0.f;
    if (q < a.n) {
        if (s_th) {                                        // rotate + translate the shared grid point
            const float gx = pre ? pre->x : a.coords[(size_t)q * 2 + 0], gy = pre ? pre->y : a.coords[(size_t)q * 2 + 1];
            xx = gx * s_th[0] - gy * s_th[1] + s_th[2];
            yy = gx * s_th[1] + gy * s_th[0] + s_th[3];
        } else {
            xx = pre ? pre->x : a.coords[((size_t)bidx * a.n + q) * 2 + 0];
            yy = pre ? pre->y : a.coords[((size_t)bidx * a.n + q) * 2 + 1];
        }
    }
    #pragma unroll
    for (int i = 0; i < G::SPT; ++i) {
        const int kg = tid / MT + i * (G::NT / MT);
        float4 h = make_float4(0, 0, 0, 0);
        if (q < a.n) {
            if (s_xy && kg == 0) { s_xy[2 * p] = xx; s_xy[2 * p + 1] = yy; }
            const int f = kg * 4;
            const float4 w0 = amx_ld4(wc + 2 * f), w1 = amx_ld4(wc + 2 * f + 4);       // Wc[f..f+3][0..1]
            const float4 zc = amx_ld4(s_zc + f);
            h.x = fmaf(w0.x, xx, fmaf(w0.y, yy, zc.x));
            h.y = fmaf(w0.z, xx, fmaf(w0.w, yy, zc.y));
            h.z = fmaf(w1.x, xx, fmaf(w1.y, yy, zc.z));
            h.w = fmaf(w1.z, xx, fmaf(w1.w, yy, zc.w));
            if (!a.skip) { h.x = rd_tanh(h.x); h.y = rd_tanh(h.y); h.z = rd_tanh(h.z); h.w = rd_tanh(h.w); }
        }
        amx_st4(dst + ((size_t)kg * G::PS + p) * 4, h);
        if (gsave) amx_st4(gsave + ((size_t)kg * a.npad + p) * 4, h);
    }
}

// (x', y') of the tile's pixels only (saved mode of the backward kernel: h0 itself comes from HBM)
template <int MT>
__device__ __forceinline__ void coord_xy(const RDecArgs& a, int bidx, int pix0, float* s_xy, const float* s_th, int tid) {
    if (tid < MT) {
        const int q = pix0 + tid;
        float xx = 0.f, yy = 0.f;
        if (q < a.n) {
            if (s_th) {
                const float gx = a.coords[(size_t)q * 2 + 0], gy = a.coords[(size_t)q * 2 + 1];
                xx = gx * s_th[0] - gy * s_th[1] + s_th[2];
                yy = gx * s_th[1] + gy * s_th[0] + s_th[3];
            } else {
                xx = a.coords[((size_t)bidx * a.n + q) * 2 + 0];
                yy = a.coords[((size_t)bidx * a.n + q) * 2 + 1];
            }
        }
        s_xy[2 * tid] = xx; s_xy[2 * tid + 1] = yy;
    }
}

// One hidden layer on a tile: dst = tanh(W src + b) [+ res].  src/dst/res are KG-layout LDS images.
// W fragments of one layer for this lane: rows 16 wave + p, k-groups 4 c + g
template <int HID>
__device__ __forceinline__ void load_wfrag(const float* Wl, int wave, int lane, float4* areg) {
    const int p = lane & 15, g = lane >> 4;
    #pragma unroll
    for (int c = 0; c < HID / 16; ++c) areg[c] = amx_ld4(Wl + (size_t)(16 * wave + p) * HID + 16 * c + 4 * g);
}

template <int HID> struct WFrag { float4 v[HID / 16]; };

// PIPE (round 5, forward kernel): the caller hands over this layer's weight fragments in `wio` — fetched while the PREVIOUS
// layer ran its epilogue — and gets the NEXT layer's (`Wnext`) back in the same registers: the loads are issued right
// after the last MFMA, when the fragments are dead, and fly during the tanh epilogue, the barrier and (for the wrap to the
// next tile) the output / coordinate layers.  Without it every layer of every tile started with an L2 round trip that
// nothing hid; keeping all layers' fragments resident instead costs 32 registers per layer and the second workgroup per CU.
template <int HID, int MT, bool PIPE = false>
__device__ __forceinline__ void hidden_layer(const float* Wl, const float* bl, const float* src, float* dst,
                                             const float* res, int wave, int lane, float* gsave = nullptr,
                                             int npad = 0, WFrag<HID>* wio = nullptr, const float* Wnext = nullptr) {
    using G = Geo<HID, MT>;
    const int p = lane & 15, g = lane >> 4;
    float4 areg[HID / 16];
    if constexpr (PIPE) {
        #pragma unroll
        for (int c = 0; c < HID / 16; ++c) areg[c] = wio->v[c];
    } else {
        load_wfrag<HID>(Wl, wave, lane, areg);
    }
    f32x4 acc[G::PT];
    #pragma unroll
    for (int t = 0; t < G::PT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
    for (int c = 0; c < HID / 16; ++c) {
        float4 bq[G::PT];
        #pragma unroll
        for (int t = 0; t < G::PT; ++t) bq[t] = amx_ld4(src + ((size_t)(4 * c + g) * G::PS + 16 * t + p) * 4);
        // k-subgroup outermost so that consecutive MFMAs target different accumulators (40-cycle dependent
        // latency vs 32-cycle issue of v_mfma_f32_16x16x4_f32)
        #pragma unroll
        for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].x, bq[t].x, acc[t], 0, 0, 0);
        #pragma unroll
        for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].y, bq[t].y, acc[t], 0, 0, 0);
        #pragma unroll
        for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].z, bq[t].z, acc[t], 0, 0, 0);
        #pragma unroll
        for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].w, bq[t].w, acc[t], 0, 0, 0);
    }
    if constexpr (PIPE) load_wfrag<HID>(Wnext, wave, lane, wio->v);
    // D'[row = feature 16*wave + 4g + r][col = pixel 16t + p]
    const float4 bias = amx_ld4(bl + 16 * wave + 4 * g);
    #pragma unroll
    for (int t = 0; t < G::PT; ++t) {
        float4 v;
        v.x = rd_tanh(acc[t][0] + bias.x); v.y = rd_tanh(acc[t][1] + bias.y);
        v.z = rd_tanh(acc[t][2] + bias.z); v.w = rd_tanh(acc[t][3] + bias.w);
        const size_t o = ((size_t)(4 * wave + g) * G::PS + 16 * t + p) * 4;
        if (res) { const float4 r = amx_ld4(res + o); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        amx_st4(dst + o, v);
        // kept for backward: the same [feature/4][pixel][4] image in HBM (16 lanes = 256 contiguous bytes per plane)
        if (gsave) amx_st4(gsave + ((size_t)(4 * wave + g) * npad + 16 * t + p) * 4, v);
    }
}

// out[p][c] = Wo[c] . h[p] + bo[c] for the tile; result left in s_out[c * MT + p]
template <int HID, int MT>
__device__ __forceinline__ void output_layer(const RDecArgs& a, const float* h, float* s_part, float* s_out,
                                             int tid, const float* wo = nullptr) {
    using G = Geo<HID, MT>;
    if (!wo) wo = a.Wo;                          // (the kernels pass their LDS copy: s_wo)
    const int p = tid % MT, part = tid / MT;
    for (int c = 0; c < a.C; ++c) {
        float acc = 0.f;
        for (int kg = part; kg < G::KG; kg += G::TPP) {
            const float4 v = amx_ld4(h + ((size_t)kg * G::PS + p) * 4);
            const float4 w = amx_ld4(wo + (size_t)c * HID + kg * 4);
            acc = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, acc))));
        }
        s_part[part * MT + p] = acc;
        __syncthreads();
        if (tid < MT) {
            float t = a.bo[c];
            #pragma unroll
            for (int q = 0; q < G::TPP; ++q) t += s_part[q * MT + tid];
            s_out[c * MT + tid] = t;
        }
        __syncthreads();
    }
}

// zc[f] = bc[f] + sum_l Wz[f][l] z[l]
template <int HID>
__device__ __forceinline__ void latent_bias(const RDecArgs& a, int bidx, float* s_zc, float* s_z, float* s_th,
                                            int tid, float* s_wc = nullptr, float* s_wo = nullptr) {
    // the coordinate layer's and the output layer's weights (HID x 2, C x HID) are read by every thread in every tile:
    // one LDS copy per workgroup instead of global loads in the tile loop (round 3, profiles/r03_rdecoder_phases.log)
    if (s_wc) for (int i = tid; i < 2 * HID; i += 4 * HID) s_wc[i] = a.Wc[i];
    if (s_wo) for (int i = tid; i < a.C * HID; i += 4 * HID) s_wo[i] = a.Wo[i];
    if (tid < a.L) s_z[tid] = a.z[(size_t)bidx * a.L + tid];
    if (a.theta && tid == 0) {                   // (cos phi, sin phi, dx, dy) of this sample
        const float phi = a.theta[(size_t)bidx * 3];
        s_th[0] = cosf(phi); s_th[1] = sinf(phi);
        s_th[2] = a.theta[(size_t)bidx * 3 + 1]; s_th[3] = a.theta[(size_t)bidx * 3 + 2];
    }
    __syncthreads();
    if (tid < HID) {
        float v = a.bc[tid];
        for (int l = 0; l < a.L; ++l) v = fmaf(a.Wz[tid * a.L + l], s_z[l], v);
        s_zc[tid] = v;
    }
    __syncthreads();
}

// --------------------------------------------------------------------------------------------------
template <int HID, int MT>
__global__ __launch_bounds__(4 * HID) void rdecoder_fwd_kernel(RDecArgs a) {
    using G = Geo<HID, MT>;
    AMX_DYN_SMEM(float, smem);
    float* buf0 = smem;
    float* buf1 = buf0 + G::BUF;
    float* buf2 = buf1 + G::BUF;                      // only with skip (keeps h0)
    float* s_zc = buf1 + G::BUF * (a.skip ? 2 : 1);
    float* s_z = s_zc + HID;
    float* s_part = s_z + MAXL;
    float* s_out = s_part + G::TPP * MT;              // [MAXC][MT]
    float* s_th = s_out + MAXC * MT;                  // [4]
    float* s_wc = s_th + 4;                           // [HID][2]  (16-byte aligned: every size above is a multiple of 4 floats)
    float* s_wo = s_wc + 2 * HID;                     // [MAXC][HID]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bidx = blockIdx.x;
#if RD_FWD_ASYM && !defined(AMX_EMU)
    // Two workgroups share a CU (106 registers, 75 KB of LDS each) and run the SAME phase sequence from the same start: their
    // VALU phases (coordinate layer, tanh epilogues, output layer) coincide and the matrix pipe idles under both, then both
    // want it at once.  Lock step is an attractor with fair arbitration (the workgroup that falls behind gets the pipe to
    // itself and catches up).  RD_FWD_ASYM makes the pair asymmetric — the workgroup in the CU's SECOND LDS slot (non-zero
    // LDS base in HW_REG_LDS_ALLOC) either runs at raised wave priority (1: it takes the pipe whenever it wants it, the other
    // one fills its VALU gaps) or starts half a tile late (2) — so that one's matrix phase falls into the other's VALU phase.
    {
        unsigned la;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
        if ((la & 0xfffu) != 0) {
            if (RD_FWD_ASYM == 1) __builtin_amdgcn_s_setprio(2);
            else for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(100);
        }
    }
#endif
    latent_bias<HID>(a, bidx, s_zc, s_z, s_th, tid, s_wc, s_wo);
    const float* th = a.theta ? s_th : nullptr;
    WFrag<HID> wf;
    if (RD_FWD_WPIPE) load_wfrag<HID>(a.W, wave, lane, wf.v);      // layer 0's fragments for the first tile
    RawXY xy_cur = load_raw_xy<MT>(a, bidx, 0, tid, th != nullptr);
    for (int pix0 = 0; pix0 < a.n; pix0 += MT) {
        float* h0 = a.skip ? buf2 : buf0;
        RawXY xy_nxt = xy_cur;
        if (RD_COORD_PREFETCH && pix0 + MT < a.n) xy_nxt = load_raw_xy<MT>(a, bidx, pix0 + MT, tid, th != nullptr);
        coord_layer<HID, MT>(a, bidx, pix0, s_zc, h0, nullptr, th, tid,
                             (RD_SAVE_H0 && a.hsave) ? a.hsave + ((size_t)bidx * RD_PLANES(a.NL) * G::KG * a.npad + pix0) * 4 : nullptr,
                             s_wc, RD_COORD_PREFETCH ? &xy_cur : nullptr);
        xy_cur = xy_nxt;
        __syncthreads();
        const float* src = h0;
        for (int l = 0; l < a.NL; ++l) {
            float* dst = (src == buf0) ? buf1 : buf0;
            float* gs = a.hsave ? a.hsave + (((size_t)bidx * RD_PLANES(a.NL) + l + (RD_SAVE_H0 ? 1 : 0)) * G::KG * a.npad + pix0) * 4 : nullptr;
            if (RD_FWD_WPIPE)
                hidden_layer<HID, MT, true>(nullptr, a.b + (size_t)l * HID, src, dst, a.skip ? h0 : nullptr, wave, lane, gs,
                                            a.npad, &wf, a.W + (size_t)(l + 1 < a.NL ? l + 1 : 0) * HID * HID);
            else
                hidden_layer<HID, MT>(a.W + (size_t)l * HID * HID, a.b + (size_t)l * HID, src, dst,
                                      a.skip ? h0 : nullptr, wave, lane, gs, a.npad);
            __syncthreads();
            src = dst;
        }
        output_layer<HID, MT>(a, src, s_part, s_out, tid, s_wo);
        for (int e = tid; e < MT * a.C; e += G::NT) {                 // xrec[b][pixel][channel]
            const int pp = e / a.C, c = e - pp * a.C;
            if (pix0 + pp < a.n) a.xrec[((size_t)bidx * a.n + pix0 + pp) * a.C + c] = s_out[c * MT + pp];
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
// CM: compile-time bound of the per-channel register accumulators (1 for grey-scale patches, MAXC otherwise) — with MAXC for
// everything the six unused accumulators of the one-channel case sat in registers of a kernel at its 256-register limit
template <int HID, int MT, int NL, bool SAVED, int CM = MAXC>
__global__ __launch_bounds__(4 * HID) void rdecoder_bwd_kernel(RDecArgs a) {
    using G = Geo<HID, MT>;
    AMX_DYN_SMEM(float, smem);
    float* H[NL + 1];
    #pragma unroll
    for (int l = 0; l <= NL; ++l) H[l] = smem + (size_t)l * G::BUF;
    float* GR = smem + (size_t)(NL + 1) * G::BUF;                     // skip only: residual gradient
    float* s_zc = smem + (size_t)(NL + 1 + (a.skip ? 1 : 0)) * G::BUF;
    float* s_z = s_zc + HID;
    float* s_part = s_z + MAXL;
    float* s_out = s_part + G::TPP * MT;           // dout [MAXC][MT]
    float* s_xy = s_out + MAXC * MT;               // [MT][2]
    float* s_red = s_xy + 2 * MT;                  // [NT] float4 scratch for the final reductions
    float* s_th = s_red + 4 * G::NT;               // [4]
    float* s_wc = s_th + 4;                        // [HID][2]
    float* s_wo = s_wc + 2 * HID;                  // [MAXC][HID]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int bidx = blockIdx.x;
    latent_bias<HID>(a, bidx, s_zc, s_z, s_th, tid, s_wc, s_wo);
    const float* th = a.theta ? s_th : nullptr;
    float a_phi = 0.f, a_tr = 0.f;                 // theta mode: thread (pixel slot, component) partial sums

    // persistent accumulators
    f32x4 accW[NL][HID / 16];
    float accb[NL];
    #pragma unroll
    for (int l = 0; l < NL; ++l) {
        accb[l] = 0.f;
        #pragma unroll
        for (int c = 0; c < HID / 16; ++c) accW[l][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // per-feature accumulators: thread tid owns feature fq for the qq-th quarter of every tile's pixels
    const int fq = tid & (HID - 1), qq = tid / HID;
    float aWo[CM], aWc0 = 0.f, aWc1 = 0.f, aZc = 0.f;
    float abo[CM];
    #pragma unroll
    for (int c = 0; c < CM; ++c) { aWo[c] = 0.f; abo[c] = 0.f; }

#ifdef AMX_RDEC_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pl = __builtin_amdgcn_s_memtime();
    const unsigned long long pstart = pl;
#endif
    // Saved-activation mode (a.hsaved): the forward kernel kept h_1..h_NL in HBM; a tile's images are fetched into
    // registers ahead of use (SPT float4 per layer and thread, 1 KB contiguous per wave and plane) and dropped into LDS
    // where the recompute would have written them — no forward MFMAs, no tanh epilogues in this kernel.
    constexpr int NPL = SAVED ? RD_PLANES(NL) : 1;               // planes fetched per tile
    constexpr int P0 = RD_SAVE_H0 ? 0 : 1;                       // H[] image the first plane belongs to
    float4 pre[NPL][G::SPT];
    const float* hs = SAVED ? a.hsaved + (size_t)bidx * RD_PLANES(NL) * G::KG * a.npad * 4 : nullptr;
    // a thread's slots are (kg0 + i * NT / MT, pp0): one per-thread base per tile, wave-uniform strides between its loads
    const int kg0 = tid / MT, pp0 = tid - kg0 * MT;
    auto fetch = [&](int pix0) {
        if (!SAVED) return;
        const float* hb = hs + ((size_t)kg0 * a.npad + pix0 + pp0) * 4;
        #pragma unroll
        for (int l = 0; l < NPL; ++l)
            #pragma unroll
            for (int i = 0; i < G::SPT; ++i)
                pre[l][i] = amx_ld4(hb + (size_t)(l * G::KG + i * (G::NT / MT)) * a.npad * 4);
    };
    if (SAVED) fetch(0);
    RawXY bxy = load_raw_xy<MT>(a, bidx, 0, tid, th != nullptr);          // this tile's raw coordinate pair (prefetched below)
    for (int pix0 = 0; pix0 < a.n; pix0 += MT) {
        // ---- recompute forward for the tile
        if (SAVED && RD_SAVE_H0) coord_xy<MT>(a, bidx, pix0, s_xy, th, tid);
        else coord_layer<HID, MT>(a, bidx, pix0, s_zc, H[0], s_xy, th, tid, nullptr, s_wc, RD_COORD_PREFETCH_BWD ? &bxy : nullptr);
        if (SAVED) {
            #pragma unroll
            for (int l = 0; l < NPL; ++l)
                #pragma unroll
                for (int i = 0; i < G::SPT; ++i) {
                    const int s = tid + i * G::NT;
                    const int kg = s / MT, pp = s - kg * MT;
                    amx_st4(H[l + P0] + ((size_t)kg * G::PS + pp) * 4, pre[l][i]);
                }
            // the tile's upstream gradient goes to LDS under the same barrier (it had one of its own)
            for (int e = tid; e < MT * a.C; e += G::NT) {
                const int pp = e / a.C, c = e - pp * a.C;
                const int q = pix0 + pp;
                s_out[c * MT + pp] = q < a.n ? a.dxrec[((size_t)bidx * a.n + q) * a.C + c] : 0.f;
            }
            __syncthreads();
            RD_TICK(0);
        } else {
            __syncthreads();
            RD_TICK(0);
            #pragma unroll
            for (int l = 0; l < NL; ++l) {
                hidden_layer<HID, MT>(a.W + (size_t)l * HID * HID, a.b + (size_t)l * HID, H[l], H[l + 1],
                                      a.skip ? H[0] : nullptr, wave, lane);
                __syncthreads();
            }
        }
        RD_TICK(1);
        // ---- output layer backward: dout, dWo, dbo, ga_NL (in place over H[NL])
        if (!SAVED) {
            for (int e = tid; e < MT * a.C; e += G::NT) {
                const int pp = e / a.C, c = e - pp * a.C;
                const int q = pix0 + pp;
                s_out[c * MT + pp] = q < a.n ? a.dxrec[((size_t)bidx * a.n + q) * a.C + c] : 0.f;
            }
            __syncthreads();
        }
        #pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c >= a.C) break;
            if (tid < MT) abo[c] += s_out[c * MT + tid];
            {                                    // dWo[c][f] += sum_p dout[p][c] * h_NL[p][f]: feature fq, the qq-th
                // quarter of the tile's pixels (all 4 * HID threads work; the quarters are summed once, after the tile loop)
                const float* hf = H[NL] + (size_t)(fq >> 2) * G::PS * 4 + (fq & 3);
                float t = aWo[c];
                #pragma unroll 8
                for (int pp = qq * (MT / 4); pp < (qq + 1) * (MT / 4); ++pp) t = fmaf(s_out[c * MT + pp], hf[pp * 4], t);
                aWo[c] = t;
            }
        }
        __syncthreads();
        #pragma unroll
        for (int i = 0; i < G::SPT; ++i) {
            const int s = tid + i * G::NT;
            const int kg = s / MT, pp = s - kg * MT;
            const size_t so = ((size_t)kg * G::PS + pp) * 4;         // this slot in an activation image
            float4 h = amx_ld4(H[NL] + so);
            float4 gh = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c = 0; c < a.C; ++c) {
                const float d = s_out[c * MT + pp];
                const float4 w = amx_ld4(s_wo + (size_t)c * HID + kg * 4);
                gh.x = fmaf(d, w.x, gh.x); gh.y = fmaf(d, w.y, gh.y); gh.z = fmaf(d, w.z, gh.z); gh.w = fmaf(d, w.w, gh.w);
            }
            if (a.skip) {
                const float4 r = amx_ld4(H[0] + so);
                amx_st4(GR + so, gh);                                   // g_res = gh_NL
                h.x -= r.x; h.y -= r.y; h.z -= r.z; h.w -= r.w;         // t = h - residual
            }
            gh.x *= 1.f - h.x * h.x; gh.y *= 1.f - h.y * h.y; gh.z *= 1.f - h.z * h.z; gh.w *= 1.f - h.w * h.w;
            amx_st4(H[NL] + so, gh);
        }
        __syncthreads();
        RD_TICK(2);
        // ---- hidden layers, last to first
        #pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            const float* ga = H[l + 1];          // gradient w.r.t. the pre-activation of layer l
            const float* hin = H[l];             // that layer's input activations
            // wgrad: dW[f][k] += sum_pix ga[pix][f] * hin[pix][k];  A[i=f][kd=pix], B[kd=pix][j=k]
            #pragma unroll 4
            for (int s4 = 0; s4 < MT / 4; ++s4) {
                const int pix = 4 * s4 + g;
                const int f = 16 * wave + p;
                const float av = ga[((size_t)(f >> 2) * G::PS + pix) * 4 + (f & 3)];
                accb[l] += av;
                #pragma unroll
                for (int c = 0; c < HID / 16; ++c) {
                    const int k = 16 * c + p;
                    const float bv = hin[((size_t)(k >> 2) * G::PS + pix) * 4 + (k & 3)];
                    accW[l][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accW[l][c], 0, 0, 0);
                }
            }
            RD_TICK(3);
            // dgrad: gh[k][pix] = sum_f W[f][k] ga[pix][f]  (A = W^T fragments, B = ga image).  (Round 5: requesting the
            // fragments inside the weight-gradient sweep, a quarter of it ahead, spills 136 B more at the 256-register limit:
            // 5.39 vs 4.80 ms per step, profiles/r05_logs/r05_rvae_bwd_wpre_rejected.log.)
            float4 areg[HID / 16];
            const float* Wt = a.Wt + (size_t)l * HID * HID;
            #pragma unroll
            for (int c = 0; c < HID / 16; ++c) areg[c] = amx_ld4(Wt + (size_t)(16 * wave + p) * HID + 16 * c + 4 * g);
            f32x4 acc[G::PT];
            #pragma unroll
            for (int t = 0; t < G::PT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int c = 0; c < HID / 16; ++c) {
                float4 bq[G::PT];
                #pragma unroll
                for (int t = 0; t < G::PT; ++t) bq[t] = amx_ld4(ga + ((size_t)(4 * c + g) * G::PS + 16 * t + p) * 4);
                #pragma unroll
                for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].x, bq[t].x, acc[t], 0, 0, 0);
                #pragma unroll
                for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].y, bq[t].y, acc[t], 0, 0, 0);
                #pragma unroll
                for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].z, bq[t].z, acc[t], 0, 0, 0);
                #pragma unroll
                for (int t = 0; t < G::PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[c].w, bq[t].w, acc[t], 0, 0, 0);
            }
            RD_TICK(4);
            __syncthreads();                     // every wave is done reading hin = H[l] and ga
            // turn gh_{l} (grad w.r.t. h_l, the layer's INPUT) into the pre-activation gradient of the
            // layer below and store it in place over H[l]
            #pragma unroll
            for (int t = 0; t < G::PT; ++t) {
                const size_t o = ((size_t)(4 * wave + g) * G::PS + 16 * t + p) * 4;
                float4 gh = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                float4 h = amx_ld4(H[l] + o);
                if (a.skip) {
                    float4 gr = amx_ld4(GR + o);
                    if (l > 0) {
                        gr.x += gh.x; gr.y += gh.y; gr.z += gh.z; gr.w += gh.w;     // g_res += gh_l
                        amx_st4(GR + o, gr);
                        const float4 r = amx_ld4(H[0] + o);
                        h.x -= r.x; h.y -= r.y; h.z -= r.z; h.w -= r.w;
                        gh.x *= 1.f - h.x * h.x; gh.y *= 1.f - h.y * h.y;
                        gh.z *= 1.f - h.z * h.z; gh.w *= 1.f - h.w * h.w;
                    } else {                     // h0 has no tanh with skip: ga_0 = gh_0 + g_res
                        gh.x += gr.x; gh.y += gr.y; gh.z += gr.z; gh.w += gr.w;
                    }
                } else {
                    gh.x *= 1.f - h.x * h.x; gh.y *= 1.f - h.y * h.y;
                    gh.z *= 1.f - h.z * h.z; gh.w *= 1.f - h.w * h.w;
                }
                amx_st4(H[l] + o, gh);
            }
            __syncthreads();
        }
        // ---- coordinate layer backward from ga_0 = H[0]
        RD_TICK(5);
        // saved mode: the next tile's images start their way from HBM now — in flight during the coordinate-layer
        // backward and the next tile's coordinate layer, i.e. while no MFMA operand / accumulator registers are live
        // (issued a whole tile ahead they cost 30 spilled registers in the <128, 64, 2> class)
        if (SAVED && pix0 + MT < a.n) fetch(pix0 + MT);
        if (RD_COORD_PREFETCH_BWD && pix0 + MT < a.n) bxy = load_raw_xy<MT>(a, bidx, pix0 + MT, tid, th != nullptr);
        float* s_g = H[1];                       // scratch [KG][MT][2] (H[1] is free now; NL >= 1)
        #pragma unroll
        for (int i = 0; i < G::SPT; ++i) {
            const int s = tid + i * G::NT;
            const int kg = s / MT, pp = s - kg * MT;
            const float4 ga0 = amx_ld4(H[0] + ((size_t)kg * G::PS + pp) * 4);
            const bool ok = pix0 + pp < a.n;
            const int f = kg * 4;
            const float4 w0 = amx_ld4(s_wc + 2 * f), w1 = amx_ld4(s_wc + 2 * f + 4);   // Wc[f..f+3][0..1]
            float gx = ga0.x * w0.x + ga0.y * w0.z + ga0.z * w1.x + ga0.w * w1.z;
            float gy = ga0.x * w0.y + ga0.y * w0.w + ga0.z * w1.y + ga0.w * w1.w;
            s_g[((size_t)kg * MT + pp) * 2 + 0] = ok ? gx : 0.f;
            s_g[((size_t)kg * MT + pp) * 2 + 1] = ok ? gy : 0.f;
        }
        {                                        // dWc[f][:] += sum_p ga0[p][f] * (x', y');  dzc[f] += sum_p ga0
            const float* gf = H[0] + (size_t)(fq >> 2) * G::PS * 4 + (fq & 3);      // (feature fq, pixel quarter qq)
            const int np = a.n - pix0 < MT ? a.n - pix0 : MT;
            const int pe = (qq + 1) * (MT / 4) < np ? (qq + 1) * (MT / 4) : np;
            for (int pp = qq * (MT / 4); pp < pe; ++pp) {
                const float gv = gf[pp * 4];
                aWc0 = fmaf(gv, s_xy[2 * pp], aWc0);
                aWc1 = fmaf(gv, s_xy[2 * pp + 1], aWc1);
                aZc += gv;
            }
        }
        __syncthreads();
        // (d x', d y') of a pixel = sum over the feature groups: every thread sums KG / NQ groups of one (pixel, component),
        // then 2 MT threads add the NQ partial sums in order (the round-1/2 form let 2 MT threads walk all KG groups while
        // the others waited: ~2 k clocks of a 64 k-clock tile)
        constexpr int NQ = G::NT / (2 * MT) >= 1 ? G::NT / (2 * MT) : 1;
        if (NQ > 1) {
            const int idx = tid % (2 * MT), q = tid / (2 * MT);
            const int pp = idx >> 1, comp = idx & 1;
            float t = 0.f;
            #pragma unroll
            for (int kg = q * (G::KG / NQ); kg < (q + 1) * (G::KG / NQ); ++kg) t += s_g[((size_t)kg * MT + pp) * 2 + comp];
            s_red[q * 2 * MT + idx] = t;
            __syncthreads();
        }
        if (tid < 2 * MT) {
            const int pp = tid >> 1, comp = tid & 1;
            float t = 0.f;
            if (NQ > 1) {
                #pragma unroll
                for (int q = 0; q < NQ; ++q) t += s_red[q * 2 * MT + tid];
            } else {
                for (int kg = 0; kg < G::KG; ++kg) t += s_g[((size_t)kg * MT + pp) * 2 + comp];
            }
            if (pix0 + pp < a.n) {
                if (th) {
                    // d/dphi of (x', y') = (-(y' - dy), x' - dx);  d/d(dx, dy) = identity
                    a_tr += t;
                    a_phi += comp == 0 ? -t * (s_xy[2 * pp + 1] - th[3]) : t * (s_xy[2 * pp] - th[2]);
                } else {
                    a.dcoords[((size_t)bidx * a.n + pix0 + pp) * 2 + comp] = t;
                }
            }
        }
        __syncthreads();
        RD_TICK(6);
    }
#ifdef AMX_RDEC_PROFILE
    pt[7] = __builtin_amdgcn_s_memtime() - pstart;
    if (a.prof && lane == 0)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)blockIdx.x * (HID / 16) + wave) * 8 + i] = pt[i];
#endif
    if (th) {                                     // fixed-order sums over the 2*MT (pixel slot, component) threads
        __syncthreads();
        if (tid < 2 * MT) { s_red[2 * tid] = a_phi; s_red[2 * tid + 1] = a_tr; }
        __syncthreads();
        if (tid < 3) {
            float t = 0.f;
            if (tid == 0) { for (int q = 0; q < 2 * MT; ++q) t += s_red[2 * q]; }
            else { for (int q = tid - 1; q < 2 * MT; q += 2) t += s_red[2 * q + 1]; }
            a.dtheta[(size_t)bidx * 3 + tid] = t;
        }
        __syncthreads();
    }

    // ---- per-sample partial rows
    #pragma unroll
    for (int l = 0; l < NL; ++l) {
        #pragma unroll
        for (int c = 0; c < HID / 16; ++c)
            #pragma unroll
            for (int r = 0; r < 4; ++r)          // D[row = f = 16w+4g+r][col = k = 16c+p]
                a.pW[(((size_t)bidx * NL + l) * HID + 16 * wave + 4 * g + r) * HID + 16 * c + p] = accW[l][c][r];
        float sb = accb[l];
        sb += __shfl_xor(sb, 16); sb += __shfl_xor(sb, 32);
        if (g == 0) a.pb[((size_t)bidx * NL + l) * HID + 16 * wave + p] = sb;
    }
    __syncthreads();
    {   // the four pixel quarters of every per-feature accumulator, summed in quarter order
        float4* r4 = reinterpret_cast<float4*>(s_red);               // [NT] float4
        r4[tid] = make_float4(aWc0, aWc1, aZc, 0.f);
        __syncthreads();
        if (tid < HID) {
            float4 t = r4[tid];
            #pragma unroll
            for (int q = 1; q < 4; ++q) { const float4 u = r4[tid + q * HID]; t.x += u.x; t.y += u.y; t.z += u.z; }
            a.pWc[((size_t)bidx * HID + tid) * 2 + 0] = t.x;
            a.pWc[((size_t)bidx * HID + tid) * 2 + 1] = t.y;
            a.pbc[(size_t)bidx * HID + tid] = t.z;
            s_zc[tid] = t.z;                     // dzc, consumed below for dWz / dz
        }
        __syncthreads();
        #pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c >= a.C) break;
            s_red[tid] = aWo[c];
            __syncthreads();
            if (tid < HID) a.pWo[((size_t)bidx * a.C + c) * HID + tid] = (s_red[tid] + s_red[tid + HID]) + (s_red[tid + 2 * HID] + s_red[tid + 3 * HID]);
            __syncthreads();
        }
    }
    // dbo
    __syncthreads();
    #pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c >= a.C) break;
        s_red[tid] = tid < MT ? abo[c] : 0.f;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int q = 0; q < MT; ++q) t += s_red[q]; a.pbo[(size_t)bidx * a.C + c] = t; }
        __syncthreads();
    }
    // dWz[f][l] = dzc[f] * z[l];  dz[l] = sum_f Wz[f][l] dzc[f]     (s_zc now holds dzc)
    __syncthreads();
    if (tid < HID)
        for (int l = 0; l < a.L; ++l) a.pWz[((size_t)bidx * HID + tid) * a.L + l] = s_zc[tid] * s_z[l];
    if (tid < a.L) {
        float t = 0.f;
        for (int f = 0; f < HID; ++f) t = fmaf(a.Wz[f * a.L + tid], s_zc[f], t);
        a.dz[(size_t)bidx * a.L + tid] = t;
    }
}

// --------------------------------------------------------------------------------------------------
template <int HID, int MT>
static size_t fwd_lds(int skip) {
    using G = Geo<HID, MT>;
    return ((size_t)G::BUF * (skip ? 3 : 2) + HID + MAXL + G::TPP * MT + MAXC * MT + 4 + 2 * HID + MAXC * HID) * sizeof(float);
}
template <int HID, int MT, int NL>
static size_t bwd_lds(int skip) {
    using G = Geo<HID, MT>;
    return ((size_t)G::BUF * (NL + 1 + (skip ? 1 : 0)) + HID + MAXL + G::TPP * MT + MAXC * MT + 2 * MT + 4 * G::NT + 4 + 2 * HID +
            MAXC * HID) * sizeof(float);
}

template <int HID, int MT>
static int launch_fwd(const RDecArgs& a, hipStream_t s) {
    const size_t lds = fwd_lds<HID, MT>(a.skip);
    if (lds > 160 * 1024) AMX_BADARG(20);
    AMX_ALLOW_160K_LDS(rdecoder_fwd_kernel<HID, MT>);
    AMX_LAUNCH((rdecoder_fwd_kernel<HID, MT>), dim3(a.B), dim3(4 * HID), lds, s, a);
    AMX_CHECK_LAUNCH();
    return 0;
}

template <int HID, int MT, int NL, bool SAVED>
static int launch_bwd_v(const RDecArgs& a, hipStream_t s) {
    const size_t lds = bwd_lds<HID, MT, NL>(a.skip);
    if (lds > 160 * 1024) AMX_BADARG(20);
    if (a.C == 1 && RD_BWD_C1) {                     // grey-scale patches (the reference's default): one-channel accumulators
        AMX_ALLOW_160K_LDS(rdecoder_bwd_kernel<HID, MT, NL, SAVED, 1>);
        AMX_LAUNCH((rdecoder_bwd_kernel<HID, MT, NL, SAVED, 1>), dim3(a.B), dim3(4 * HID), lds, s, a);
    } else {
        AMX_ALLOW_160K_LDS(rdecoder_bwd_kernel<HID, MT, NL, SAVED>);
        AMX_LAUNCH((rdecoder_bwd_kernel<HID, MT, NL, SAVED>), dim3(a.B), dim3(4 * HID), lds, s, a);
    }
    AMX_CHECK_LAUNCH();
    return 0;
}

template <int HID, int MT, int NL>
static int launch_bwd(const RDecArgs& a, hipStream_t s) {
    return a.hsaved ? launch_bwd_v<HID, MT, NL, true>(a, s) : launch_bwd_v<HID, MT, NL, false>(a, s);
}

static int check_common(const RDecArgs& a, int hid) {
    if (!a.coords || !a.z || !a.Wc || !a.bc || !a.Wz || !a.W || !a.b || !a.Wo || !a.bo) AMX_BADARG(1);
    if (a.B <= 0 || a.n <= 0 || a.L < 1 || a.L > MAXL || a.NL < 1 || a.NL > 5) AMX_BADARG(2);
    if (hid != 32 && hid != 64 && hid != 128) AMX_BADARG(3);
    if (a.C < 1 || a.C > MAXC) AMX_BADARG(5);
    return 0;
}

extern "C" long amx_rdecoder_hsave_floats(int B, int n, int hid, int NL) {
    if (B <= 0 || n <= 0 || NL < 1 || (hid != 32 && hid != 64 && hid != 128)) return -1;
    const long npad = ((long)n + 127) / 128 * 128;
    return (long)B * RD_PLANES(NL) * hid * npad;
}

extern "C" int amx_rdecoder_fwd_save(const float* coords, const float* theta, const float* z, const float* Wc,
                                     const float* bc, const float* Wz, const float* W, const float* b, const float* Wo,
                                     const float* bo, float* xrec, float* hsave, int B, int n, int L, int hid, int NL,
                                     int skip, int C, void* stream);

extern "C" int amx_rdecoder_fwd(const float* coords, const float* theta, const float* z, const float* Wc,
                                const float* bc, const float* Wz, const float* W, const float* b, const float* Wo,
                                const float* bo, float* xrec, int B, int n, int L, int hid, int NL, int skip,
                                int C, void* stream) {
    return amx_rdecoder_fwd_save(coords, theta, z, Wc, bc, Wz, W, b, Wo, bo, xrec, nullptr, B, n, L, hid, NL, skip, C,
                                 stream);
}

extern "C" int amx_rdecoder_fwd_save(const float* coords, const float* theta, const float* z, const float* Wc,
                                     const float* bc, const float* Wz, const float* W, const float* b, const float* Wo,
                                     const float* bo, float* xrec, float* hsave, int B, int n, int L, int hid, int NL,
                                     int skip, int C, void* stream) {
    RDecArgs a = {};
    a.hsave = hsave; a.npad = (n + 127) / 128 * 128;
    a.coords = coords; a.theta = theta; a.z = z; a.Wc = Wc; a.bc = bc; a.Wz = Wz; a.W = W; a.b = b; a.Wo = Wo;
    a.bo = bo; a.xrec = xrec; a.B = B; a.n = n; a.L = L; a.NL = NL; a.skip = skip; a.C = C;
    const int rc = check_common(a, hid);
    if (rc) return rc;
    if (!xrec) AMX_BADARG(4);
    hipStream_t s = (hipStream_t)stream;
    int mt = 64;      // 64-pixel tiles: 66 KB of LDS -> two workgroups per CU (58.8 % vs 53.4 % of MFMA peak measured; round 3:
                      // 32-pixel tiles, three workgroups per CU: rVAE step 4.99 -> 5.13 ms, not instantiated)
    mt = amx_knobs().rdec_fwd_mt;
    if (hid == 32) return launch_fwd<32, 128>(a, s);
    if (hid == 64) return launch_fwd<64, 128>(a, s);
    return (skip || mt == 64) ? launch_fwd<128, 64>(a, s) : launch_fwd<128, 128>(a, s);
}

extern "C" int amx_rdecoder_bwd_saved(const float* coords, const float* theta, const float* z, const float* Wc,
                                      const float* bc, const float* Wz, const float* W, const float* Wt, const float* b,
                                      const float* Wo, const float* bo, const float* dxrec, const float* hsaved,
                                      float* dcoords, float* dtheta, float* dz, float* pW, float* pb, float* pWo,
                                      float* pbo, float* pWc, float* pbc, float* pWz, int B, int n, int L, int hid,
                                      int NL, int skip, int C, void* stream);

extern "C" int amx_rdecoder_bwd(const float* coords, const float* theta, const float* z, const float* Wc,
                                const float* bc, const float* Wz, const float* W, const float* Wt, const float* b,
                                const float* Wo, const float* bo, const float* dxrec, float* dcoords,
                                float* dtheta, float* dz, float* pW, float* pb, float* pWo, float* pbo, float* pWc,
                                float* pbc, float* pWz, int B, int n, int L, int hid, int NL, int skip, int C,
                                void* stream) {
    return amx_rdecoder_bwd_saved(coords, theta, z, Wc, bc, Wz, W, Wt, b, Wo, bo, dxrec, nullptr, dcoords, dtheta, dz,
                                  pW, pb, pWo, pbo, pWc, pbc, pWz, B, n, L, hid, NL, skip, C, stream);
}

extern "C" int amx_rdecoder_bwd_saved(const float* coords, const float* theta, const float* z, const float* Wc,
                                      const float* bc, const float* Wz, const float* W, const float* Wt, const float* b,
                                      const float* Wo, const float* bo, const float* dxrec, const float* hsaved,
                                      float* dcoords, float* dtheta, float* dz, float* pW, float* pb, float* pWo,
                                      float* pbo, float* pWc, float* pbc, float* pWz, int B, int n, int L, int hid,
                                      int NL, int skip, int C, void* stream) {
    RDecArgs a = {};
    a.hsaved = hsaved; a.npad = (n + 127) / 128 * 128;
    a.coords = coords; a.theta = theta; a.z = z; a.Wc = Wc; a.bc = bc; a.Wz = Wz; a.W = W; a.Wt = Wt; a.b = b;
    a.Wo = Wo; a.bo = bo; a.dxrec = dxrec; a.dcoords = dcoords; a.dtheta = dtheta; a.dz = dz;
    a.pW = pW; a.pb = pb; a.pWo = pWo; a.pbo = pbo; a.pWc = pWc; a.pbc = pbc; a.pWz = pWz;
    a.B = B; a.n = n; a.L = L; a.NL = NL; a.skip = skip; a.C = C;
#ifdef AMX_RDEC_PROFILE
    a.prof = (unsigned long long*)amx_rdec_profile_buffer;
#endif
    const int rc = check_common(a, hid);
    if (rc) return rc;
    if (!Wt || !dxrec || !dz || !pW || !pb || !pWo || !pbo || !pWc || !pbc || !pWz) AMX_BADARG(4);
    if (theta ? !dtheta : !dcoords) AMX_BADARG(6);
    hipStream_t s = (hipStream_t)stream;
    // LDS holds NL + 1 (+1 with skip) activation images of HID x MT floats: the tile shrinks as the decoder deepens
#define RD_BWD(HID_, MT_)                                               \
    switch (NL) {                                                       \
        case 1: return launch_bwd<HID_, MT_, 1>(a, s);                  \
        case 2: return launch_bwd<HID_, MT_, 2>(a, s);                  \
        case 3: return launch_bwd<HID_, MT_, 3>(a, s);                  \
        case 4: return launch_bwd<HID_, MT_, 4>(a, s);                  \
        default: return launch_bwd<HID_, MT_, 5>(a, s);                 \
    }
    if (hid == 32) { RD_BWD(32, 64) }
    if (hid == 64) { RD_BWD(64, 64) }
    const int mt = amx_knobs().rdec_bwd_mt;
    if (mt == 64 && NL + 1 + (skip ? 1 : 0) <= 4) { RD_BWD(128, 64) }
    RD_BWD(128, 32)
#undef RD_BWD
}
