// Micro-benchmark (not part of the product library): sustained v_mfma_f32_16x16x4_f32 rate of the whole chip,
// i.e. the practical fp32-MFMA ceiling at the clock the part actually holds under matrix load.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

extern "C" double mfma_peak_tflops(int wgs, int iters, int reps) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double flops = (double)reps * wgs * 4.0 /*waves*/ * iters * 8.0 * 2048.0;
    return flops / (ms * 1e-3) / 1e12;
}
