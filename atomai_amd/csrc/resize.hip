// resize.hip — F.interpolate(x, size=(H, W), mode) of small-channel score maps, written straight into a channel
// slice of the concatenated tensor (ResHedNet.forward, atomai/nets/fcnn.py:283-295: three nb_classes-channel side
// outputs, two of them interpolated x2 / x4 to the input size, torch.cat, 1x1 conv).  Arbitrary in/out sizes with
// ATen's align_corners=False coordinate rule
//     scale = in / out (float);  src = scale * (dst + 0.5) - 0.5, clamped at 0;  i0 = floor(src), i1 = min(i0+1, in-1)
// (nearest: i = min(floor(dst * scale), in-1)).  The producing layer's pending BatchNorm affine is applied on load.
// Channel counts are tiny (<= 3 classes per source), so these are scalar per-(pixel, channel) kernels; the backward is
// a deterministic gather (ATen's CUDA backward scatters with float atomics).
#include "amx_device.h"

__device__ __forceinline__ void rs_taps(int o, int in, int out, int mode, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)in / (float)out;
    if (mode == 1) {
        int i = (int)floorf((float)o * scale);
        i0 = i1 = i < in - 1 ? i : in - 1;
        l0 = 1.f; l1 = 0.f;
        return;
    }
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void resize_cat_fwd_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int N, int h, int w, int Cs,
                                                             int C, float* __restrict__ dst, int H, int W, int Cd, int coff,
                                                             int mode) {
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); const int n = (int)(r / H);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        rs_taps(y, h, H, mode, y0, y1, ly0, ly1);
        rs_taps(x, w, W, mode, x0, x1, lx0, lx1);
        const float* b = src + (size_t)n * h * w * Cs + c;
        const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
        const float a00 = fmaf(b[((size_t)y0 * w + x0) * Cs], sc, sh);
        float v;
        if (mode == 1) v = a00;
        else {
            const float a01 = fmaf(b[((size_t)y0 * w + x1) * Cs], sc, sh);
            const float a10 = fmaf(b[((size_t)y1 * w + x0) * Cs], sc, sh);
            const float a11 = fmaf(b[((size_t)y1 * w + x1) * Cs], sc, sh);
            v = ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11);
        }
        dst[(((size_t)n * H + y) * W + x) * Cd + coff + c] = v;
    }
}

extern "C" int amx_resize_cat_fwd(const float* src, const float* scale, const float* shift, int N, int h, int w, int Cs,
                                  int C, float* dst, int H, int W, int Cd, int coff, int mode, void* stream) {
    if (!src || !dst || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) AMX_BADARG(1);
    if (C <= 0 || C > Cs || coff < 0 || coff + C > Cd || (mode != 0 && mode != 1)) AMX_BADARG(2);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(3);
    const size_t total = (size_t)N * H * W * C;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    AMX_LAUNCH(resize_cat_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, scale, shift, N, h, w, Cs, C,
               dst, H, W, Cd, coff, mode);
    AMX_CHECK_LAUNCH();
    return 0;
}

// dsrc[n][iy][ix][c] = sum over the output pixels that read (iy, ix): weight * ddst[n][oy][ox][coff + c]
// (gradient with respect to the value AFTER the producer's affine); channels c >= C of dsrc are zeroed.
__global__ __launch_bounds__(256) void resize_cat_bwd_kernel(const float* __restrict__ ddst, int N, int H, int W, int Cd,
                                                             int coff, int C, float* __restrict__ dsrc, int h, int w,
                                                             int Cs, int mode) {
    const size_t total = (size_t)N * h * w * Cs;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        size_t r = i / Cs;
        const int ix = (int)(r % w); r /= w;
        const int iy = (int)(r % h); const int n = (int)(r / h);
        float acc = 0.f;
        if (c < C) {
            // conservative candidate windows (outputs whose taps can touch this input pixel), then exact test
            int oy_lo = (int)floorf(((float)iy - 1.f) / sy) - 1, oy_hi = (int)ceilf(((float)iy + 2.f) / sy) + 1;
            int ox_lo = (int)floorf(((float)ix - 1.f) / sx) - 1, ox_hi = (int)ceilf(((float)ix + 2.f) / sx) + 1;
            if (oy_lo < 0) oy_lo = 0; if (ox_lo < 0) ox_lo = 0;
            if (oy_hi > H - 1) oy_hi = H - 1; if (ox_hi > W - 1) ox_hi = W - 1;
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int y0, y1; float ly0, ly1;
                rs_taps(oy, h, H, mode, y0, y1, ly0, ly1);
                const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
                if (wy == 0.f) continue;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    int x0, x1; float lx0, lx1;
                    rs_taps(ox, w, W, mode, x0, x1, lx0, lx1);
                    const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                    if (wx == 0.f) continue;
                    acc = fmaf(wy * wx, ddst[(((size_t)n * H + oy) * W + ox) * Cd + coff + c], acc);
                }
            }
        }
        dsrc[i] = acc;
    }
}

extern "C" int amx_resize_cat_bwd(const float* ddst, int N, int H, int W, int Cd, int coff, int C, float* dsrc, int h,
                                  int w, int Cs, int mode, void* stream) {
    if (!ddst || !dsrc || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) AMX_BADARG(1);
    if (C <= 0 || C > Cs || coff < 0 || coff + C > Cd || (mode != 0 && mode != 1)) AMX_BADARG(2);
    const size_t total = (size_t)N * h * w * Cs;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    AMX_LAUNCH(resize_cat_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ddst, N, H, W, Cd, coff, C, dsrc,
               h, w, Cs, mode);
    AMX_CHECK_LAUNCH();
    return 0;
}
