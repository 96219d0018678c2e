// adam.hip — fused flat Adam: ONE launch over the flat fp32 parameter / gradient / moment buffers
// (the reference runs torch.optim.Adam's ~60 foreach kernels, atomai/trainers/trainer.py:539,
// vitrainer.py:218).  Semantics = torch.optim.Adam defaults (amsgrad off, weight_decay 0):
//   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; p <- p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// 1-b1, 1-b2 and the bias corrections bc1 = 1-b1^t, sqrt(bc2) are formed in fp64 (as torch does) and then rounded.
// `gscale` folds the 1/world_size of the data-parallel gradient all-reduce into the same pass.
#include "amx_device.h"

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long n, float lr_over_bc1, float omb1, float b2,
                                                        float omb2, float inv_sqrt_bc2, float eps, float gscale) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pp = amx_ld4(p + i * 4), gg = amx_ld4(g + i * 4), mm = amx_ld4(m + i * 4), vv = amx_ld4(v + i * 4);
        #define AMX_ADAM1(c)                                                           \
            { const float gr = gg.c * gscale;                                          \
              mm.c = mm.c + (gr - mm.c) * omb1;                                         \
              vv.c = vv.c * b2 + gr * gr * omb2;                                        \
              pp.c -= lr_over_bc1 * (mm.c / (sqrtf(vv.c) * inv_sqrt_bc2 + eps)); }
        AMX_ADAM1(x) AMX_ADAM1(y) AMX_ADAM1(z) AMX_ADAM1(w)
        amx_st4(p + i * 4, pp); amx_st4(m + i * 4, mm); amx_st4(v + i * 4, vv);
    }
    // tail (n not a multiple of 4)
    const long t0 = n4 << 2;
    const long i = t0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gr = g[i] * gscale;
        const float mi = m[i] + (gr - m[i]) * omb1;
        const float vi = v[i] * b2 + gr * gr * omb2;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

extern "C" int amx_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, double b1,
                             double b2, float eps, double bc1, double bc2, float gscale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || bc1 <= 0.0 || bc2 <= 0.0) AMX_BADARG(1);
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) AMX_BADARG(2);
    long nb = ((n >> 2) + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    AMX_LAUNCH(adam_flat_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
               (float)((double)lr / bc1), (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(1.0 / sqrt(bc2)), eps,
               gscale);
    AMX_CHECK_LAUNCH();
    return 0;
}
