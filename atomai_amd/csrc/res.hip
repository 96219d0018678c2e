// res.hip — the two elementwise kernels of the residual block (atomai/nets/blocks.py:199-214, ResBlock.forward):
//     x = c0(x);  out = c1(x) -> bn1 -> LeakyReLU -> c2 -> bn2;  out += x;  out = LeakyReLU(out)
// The convolutions run on conv_fwd.hip / wgrad.hip (BatchNorm affine + LeakyReLU of bn1 are applied by c2's
// loader: amx_conv2d_fwd_act / amx_conv2d_wgrad_act).  What is left:
//   amx_res_out_fwd    out = LeakyReLU(t*scale + shift + r)          (bn2 affine, residual add, activation)
//   amx_lrelu_bwd      din = dout * LeakyReLU'(z),  z = ref*scale + shift   (scale == nullptr: z = ref)
//                      — the activation-after-BatchNorm backward (z = bn1 output) and the block-output backward
//                      (ref = out: LeakyReLU preserves the sign, so out > 0 <=> pre-activation > 0).
// NHWC fp32, channels padded to a multiple of 4; HBM-bound single passes, 16 B per lane.
#include "amx_device.h"

__global__ __launch_bounds__(256) void res_out_fwd_kernel(const float* __restrict__ t, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ r,
                                                          float slope, size_t n4, int G, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % G) * 4;
        float4 v = amx_ld4(t + i * 4);
        if (scale) {
            const float4 sc = amx_ld4(scale + c), sh = amx_ld4(shift + c);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        const float4 q = amx_ld4(r + i * 4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        amx_st4(out + i * 4, v);
    }
}

extern "C" int amx_res_out_fwd(const float* t, const float* scale, const float* shift, const float* r, float slope,
                               long npix, int Cs, float* out, void* stream) {
    if (!t || !r || !out || npix <= 0 || Cs <= 0 || (Cs & 3)) AMX_BADARG(1);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(2);
    const size_t n4 = (size_t)npix * (Cs / 4);
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    AMX_LAUNCH(res_out_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, scale, shift, r, slope, n4,
               Cs / 4, out);
    AMX_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ ref,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float slope, size_t n4, int G, float* __restrict__ din,
                                                        float* __restrict__ din2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % G) * 4;
        float4 z = amx_ld4(ref + i * 4);
        if (scale) {
            const float4 sc = amx_ld4(scale + c), sh = amx_ld4(shift + c);
            z.x = fmaf(z.x, sc.x, sh.x); z.y = fmaf(z.y, sc.y, sh.y);
            z.z = fmaf(z.z, sc.z, sh.z); z.w = fmaf(z.w, sc.w, sh.w);
        }
        float4 g = amx_ld4(dout + i * 4);
        g.x = z.x > 0.f ? g.x : g.x * slope; g.y = z.y > 0.f ? g.y : g.y * slope;
        g.z = z.z > 0.f ? g.z : g.z * slope; g.w = z.w > 0.f ? g.w : g.w * slope;
        amx_st4(din + i * 4, g);
        if (din2) amx_st4(din2 + i * 4, g);
    }
}

// din2 (optional): a second copy of the result, for the case where the two branches that receive this gradient
// (residual path / convolution path) must not alias.
extern "C" int amx_lrelu_bwd(const float* dout, const float* ref, const float* scale, const float* shift,
                             float slope, long npix, int Cs, float* din, float* din2, void* stream) {
    if (!dout || !ref || !din || npix <= 0 || Cs <= 0 || (Cs & 3)) AMX_BADARG(1);
    if ((scale == nullptr) != (shift == nullptr)) AMX_BADARG(2);
    const size_t n4 = (size_t)npix * (Cs / 4);
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    AMX_LAUNCH(lrelu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ref, scale, shift, slope, n4,
               Cs / 4, din, din2);
    AMX_CHECK_LAUNCH();
    return 0;
}
