// pack.hip — weight-image builders and layout converters (all HBM-bound, tiny).
//
// The public module tree keeps nn.Conv2d-shaped OIHW fp32 parameters (state-dict compatibility,
// SURVEY.md §0.4); the MFMA kernels consume a K-chunked image
//     wpk[chunk][tap][kgroup(4)][n (padded to 16)][4 channels]
// built here once per optimizer step:
//   mode 0 (forward):  k runs over the concatenated, 4-padded input channels of (src0|src1), n = cout
//   mode 1 (dgrad):    k runs over cout, n runs over the concatenated padded input channels, taps are
//                      spatially flipped (tap -> taps-1-tap)  [dX = conv(dY, flip(W)^T)]
// A PARTIAL last chunk (k space not a multiple of 16: fewer than 4 valid k-groups) is stored with the roles of k-group
// and channel-in-group swapped: slot [kg'][n][e'] holds channel (4*e' + kg') of the chunk.  The convolution kernel
// stages the matching input chunk the same way, so its ordinary fragment read (lane group g reads [g][..][0..3])
// returns channel g of k-groups 0..3 in components x..w, and only the components of VALID k-groups are issued as
// MFMAs (conv_kernel.h, TAIL).
#include "amx_device.h"

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                    int cout, int cin, int C0, int C0s, int C1, int C1s, int taps,
                                    int mode, int nop, int total, int tchunk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int t = idx;
    const int e = t & 3; t >>= 2;
    const int n = t % nop; t /= nop;
    const int kg = t & 3; t >>= 2;
    const int tap = t % taps; const int chunk = t / taps;
    // a PARTIAL last chunk is stored transposed (kgroup <-> channel-in-group), see the file header
    const int k = chunk == tchunk ? (chunk * 4 + e) * 4 + kg : (chunk * 4 + kg) * 4 + e;
    // map an index of the concatenated padded channel space to a real input channel (or -1)
    auto cat2ci = [&](int c) -> int {
        if (c < C0s) return c < C0 ? c : -1;
        c -= C0s;
        return (c < C1s && c < C1) ? C0 + c : -1;
    };
    int co, ci, tp;
    if (mode == 0) { ci = cat2ci(k); co = n < cout ? n : -1; tp = tap; }
    else { co = k < cout ? k : -1; ci = cat2ci(n); tp = taps - 1 - tap; }
    float v = 0.f;
    if (co >= 0 && ci >= 0) v = w[((size_t)co * cin + ci) * taps + tp];
    dst[idx] = v;
}

extern "C" int amx_pack_weights(const float* w_oihw, float* dst, int cout, int C0, int C0s, int C1,
                                int C1s, int taps, int mode, void* stream) {
    if (!w_oihw || !dst) AMX_BADARG(1);
    if (cout <= 0 || C0 <= 0 || C0s < C0 || C1s < C1 || (C0s & 3) || (C1s & 3)) AMX_BADARG(2);
    if (taps != 1 && taps != 9) AMX_BADARG(3);
    const int kspace = mode == 0 ? (C0s + C1s) : amx_round_up(cout, 4);
    const int nspace = mode == 0 ? cout : (C0s + C1s);
    const int nchunk = amx_ceil_div(kspace, 16);
    const int nop = amx_round_up(nspace, 16);
    const int total = nchunk * taps * 4 * nop * 4;
    AMX_LAUNCH(pack_weights_kernel, dim3(amx_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
               w_oihw, dst, cout, C0 + C1, C0, C0s, C1, C1s, taps, mode, nop, total,
               (kspace & 15) ? nchunk - 1 : -1);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Image of a convolution restricted to ONE of its sources' input channels [ci_off, ci_off + Cn) of cin_total (round 6: the
// data gradient of a layer that read a concatenation of two 32-channel sources runs as two launches of the wave-specialised
// kernel, one per source, each with the weight image of its half — engine.ConvNode._dgrad).  Same layout as
// amx_pack_weights(w, dst, cout, Cn, Cns, 0, 0, taps, mode) of the sliced tensor w[:, ci_off : ci_off + Cn].
extern "C" int amx_pack_weights_range(const float* w_oihw, float* dst, int cout, int cin_total, int ci_off, int Cn,
                                      int Cns, int taps, int mode, void* stream) {
    if (!w_oihw || !dst) AMX_BADARG(1);
    if (cout <= 0 || Cn <= 0 || Cns < Cn || (Cns & 3) || ci_off < 0 || ci_off + Cn > cin_total) AMX_BADARG(2);
    if (taps != 1 && taps != 9) AMX_BADARG(3);
    const int kspace = mode == 0 ? Cns : amx_round_up(cout, 4);
    const int nspace = mode == 0 ? cout : Cns;
    const int nchunk = amx_ceil_div(kspace, 16);
    const int nop = amx_round_up(nspace, 16);
    const int total = nchunk * taps * 4 * nop * 4;
    // (the kernel addresses w[(co * cin + ci) * taps + tap]: the row stride stays cin_total, the base moves by ci_off)
    AMX_LAUNCH(pack_weights_kernel, dim3(amx_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
               w_oihw + (size_t)ci_off * taps, dst, cout, cin_total, Cn, Cns, 0, 0, taps, mode, nop, total,
               (kspace & 15) ? nchunk - 1 : -1);
    AMX_CHECK_LAUNCH();
    return 0;
}

// Every layer's image in one launch (blockIdx.y = job): the per-layer launches sat between the convolutions of the
// forward / backward chains, 30 five-microsecond kernels per training step.
#define PACK_BATCH 8
struct PackJob { const float* w; float* dst; int cout, cin, C0, C0s, C1, C1s, taps, mode, nop, total, tchunk; };
struct PackBatch { PackJob j[PACK_BATCH]; };

__global__ void pack_weights_batch_kernel(PackBatch b) {
    const PackJob& J = b.j[blockIdx.y];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < J.total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int e = t & 3; t >>= 2;
        const int n = t % J.nop; t /= J.nop;
        const int kg = t & 3; t >>= 2;
        const int tap = t % J.taps; const int chunk = t / J.taps;
        const int k = chunk == J.tchunk ? (chunk * 4 + e) * 4 + kg : (chunk * 4 + kg) * 4 + e;
        auto cat2ci = [&](int c) -> int {
            if (c < J.C0s) return c < J.C0 ? c : -1;
            c -= J.C0s;
            return (c < J.C1s && c < J.C1) ? J.C0 + c : -1;
        };
        int co, ci, tp;
        if (J.mode == 0) { ci = cat2ci(k); co = n < J.cout ? n : -1; tp = tap; }
        else { co = k < J.cout ? k : -1; ci = cat2ci(n); tp = J.taps - 1 - tap; }
        float v = 0.f;
        if (co >= 0 && ci >= 0) v = J.w[((size_t)co * J.cin + ci) * J.taps + tp];
        J.dst[idx] = v;
    }
}

// desc: n rows of (cout, C0, C0s, C1, C1s, taps, mode); w / dst: n device pointers (host arrays of pointers)
extern "C" int amx_pack_weights_batch(const void* const* w, void* const* dst, const int* desc, int n, void* stream) {
    if (!w || !dst || !desc || n <= 0) AMX_BADARG(1);
    for (int base = 0; base < n; base += PACK_BATCH) {
        PackBatch b;
        const int m = n - base < PACK_BATCH ? n - base : PACK_BATCH;
        int maxtotal = 0;
        for (int i = 0; i < m; ++i) {
            const int* d = desc + (size_t)(base + i) * 7;
            const int cout = d[0], C0 = d[1], C0s = d[2], C1 = d[3], C1s = d[4], taps = d[5], mode = d[6];
            if (!w[base + i] || !dst[base + i]) AMX_BADARG(2);
            if (cout <= 0 || C0 <= 0 || C0s < C0 || C1s < C1 || (C0s & 3) || (C1s & 3)) AMX_BADARG(3);
            if (taps != 1 && taps != 9) AMX_BADARG(4);
            const int kspace = mode == 0 ? (C0s + C1s) : amx_round_up(cout, 4);
            const int nspace = mode == 0 ? cout : (C0s + C1s);
            PackJob& J = b.j[i];
            J.w = (const float*)w[base + i]; J.dst = (float*)dst[base + i];
            J.cout = cout; J.cin = C0 + C1; J.C0 = C0; J.C0s = C0s; J.C1 = C1; J.C1s = C1s; J.taps = taps; J.mode = mode;
            J.nop = amx_round_up(nspace, 16);
            J.total = amx_ceil_div(kspace, 16) * taps * 4 * J.nop * 4;
            J.tchunk = (kspace & 15) ? amx_ceil_div(kspace, 16) - 1 : -1;
            if (J.total > maxtotal) maxtotal = J.total;
        }
        int gx = amx_ceil_div(maxtotal, 256);
        if (gx > 256) gx = 256;
        AMX_LAUNCH(pack_weights_batch_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, b);
    }
    AMX_CHECK_LAUNCH();
    return 0;
}

// number of floats amx_pack_weights writes
extern "C" long amx_pack_weights_size(int cout, int C0s, int C1s, int taps, int mode) {
    const int kspace = mode == 0 ? (C0s + C1s) : amx_round_up(cout, 4);
    const int nspace = mode == 0 ? cout : (C0s + C1s);
    return (long)amx_ceil_div(kspace, 16) * taps * 4 * amx_round_up(nspace, 16) * 4;
}

// ---- NCHW <-> NHWC (stored channels Cs = round_up(C,4), zero padded) ----
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                    int C, int Cs, int HW) {
    const size_t total = (size_t)N * HW * Cs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        const size_t r = i / Cs;
        const size_t n = r / HW, hw = r - n * HW;
        dst[i] = c < C ? src[(n * C + c) * HW + hw] : 0.f;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                    int C, int Cs, int HW) {
    const size_t total = (size_t)N * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t hw = i % HW;
        const size_t r = i / HW;
        const size_t c = r % C, n = r / C;
        dst[i] = src[(n * HW + hw) * Cs + c];
    }
}

extern "C" int amx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int Cs, int H, int W,
                                void* stream) {
    if (!src || !dst || C <= 0 || Cs < C || (Cs & 3)) AMX_BADARG(1);
    const size_t total = (size_t)N * H * W * Cs;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    AMX_LAUNCH(nchw_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, N, C,
               Cs, H * W);
    AMX_CHECK_LAUNCH();
    return 0;
}

extern "C" int amx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int Cs, int H, int W,
                                void* stream) {
    if (!src || !dst || C <= 0 || Cs < C || (Cs & 3)) AMX_BADARG(1);
    const size_t total = (size_t)N * H * W * C;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    AMX_LAUNCH(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, N, C,
               Cs, H * W);
    AMX_CHECK_LAUNCH();
    return 0;
}

// dst += src (gradient accumulation when a kernel cannot accumulate in place)
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        float4 a = amx_ld4(dst + i * 4);
        const float4 b = amx_ld4(src + i * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        amx_st4(dst + i * 4, a);
    }
}

extern "C" int amx_add_inplace(float* dst, const float* src, long n, void* stream) {
    if (!dst || !src || n <= 0 || (n & 3)) AMX_BADARG(1);
    const size_t n4 = (size_t)n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    AMX_LAUNCH(add_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, src, n4);
    AMX_CHECK_LAUNCH();
    return 0;
}

// y = (x - sub) / div : torch_format_image's global min-max normalisation (atomai/utils/preproc.py:818-823,
// `(image_data - image_data.min()) / np.ptp(image_data)`) applied to a chunk that is already on the device.
// Two correctly rounded fp32 operations, exactly what numpy performs on a float32 stack (bit-identical result).
__global__ void sub_div_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float sub, float div) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = x[i] - sub;
        y[i] = d / div;
    }
}

extern "C" int amx_sub_div(const float* x, float* y, long n, float sub, float div, void* stream) {
    if (!x || !y || n <= 0) AMX_BADARG(1);
    const size_t nn = (size_t)n;
    const int blocks = (int)((nn + 255) / 256 < 16384 ? (nn + 255) / 256 : 16384);
    AMX_LAUNCH(sub_div_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, nn, sub, div);
    AMX_CHECK_LAUNCH();
    return 0;
}

// dst[i] = src[i], 16 B per lane, on a bounded number of workgroups.  Used by the predictor to write a chunk's
// probabilities straight into PINNED HOST memory (device-accessible under unified addressing) from a side stream:
// a kernel behind a cross-stream event wait never blocks the launching thread, whereas hipMemcpyAsync behind such a
// wait was measured to stall the host for up to two chunk times (SegPredictor.batch_predict).
__global__ __launch_bounds__(256) void copy16_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

extern "C" int amx_copy16(const void* src, void* dst, long nbytes, int max_wgs, void* stream) {
    if (!src || !dst || nbytes <= 0 || (nbytes & 15) || max_wgs <= 0) AMX_BADARG(1);
    if (((uintptr_t)src | (uintptr_t)dst) & 15) AMX_BADARG(2);
    const size_t n16 = (size_t)nbytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > (size_t)max_wgs) blocks = (size_t)max_wgs;
    AMX_LAUNCH(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
               (const float4*)src, (float4*)dst, n16);
    AMX_CHECK_LAUNCH();
    return 0;
}
