// Micro-benchmark (not part of the product library): how well does the fp32 MFMA pipe stay fed when every burst of 16
// MFMAs needs 4 ds_read_b128 first — the inner loop shape of the convolution kernel (MTW = NT = 2) — as a function of
// the number of co-resident waves per SIMD and of whether the operand reads are software-pipelined one burst ahead.
//   mode 0: no LDS at all (pure MFMA ceiling)
//   mode 1: reads issued right before the burst that uses them (what hipcc generates from the plain loop)
//   mode 2: reads of burst i+1 issued before the MFMAs of burst i (two fragment register sets)
//   mode 3: as 1, plus a __syncthreads() every 9 bursts (the chunk barrier of the convolution kernel)
//   mode 4: as 2, plus the barrier
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) smem[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* base = reinterpret_cast<const float4*>(smem) + lane;
    auto rd = [&](int i, float4* f) {
        #pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = base[((i * 4 + j) & 7) * 64];
    };
    auto burst = [&](const float4* f) {
        #define M(C) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0].C, f[2].C, acc[0], 0, 0, 0); \
                     acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0].C, f[3].C, acc[1], 0, 0, 0); \
                     acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1].C, f[2].C, acc[2], 0, 0, 0); \
                     acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1].C, f[3].C, acc[3], 0, 0, 0);
        M(x) M(y) M(z) M(w)
        #undef M
    };
    float4 f0[4], f1[4];
    if (MODE == 0) { rd(0, f0); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            #pragma unroll
            for (int t = 0; t < 9; ++t) burst(f0);
        } else if (MODE == 1 || MODE == 3) {
            #pragma unroll
            for (int t = 0; t < 9; ++t) { rd(t, f0); burst(f0); }
            if (MODE == 3) __syncthreads();
        } else {
            rd(0, f0);
            #pragma unroll
            for (int t = 0; t < 9; t += 2) {
                if (t + 1 < 9) rd(t + 1, f1);
                __builtin_amdgcn_sched_barrier(0);
                burst(f0);
                if (t + 1 < 9) {
                    if (t + 2 < 9) rd(t + 2, f0);
                    __builtin_amdgcn_sched_barrier(0);
                    burst(f1);
                }
            }
            if (MODE == 4) __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static double run(int wgs_per_cu, int iters, int reps) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // LDS per workgroup sized so that exactly `wgs_per_cu` workgroups fit a CU (160 KB)
    const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) / 1024 * 1024 - (wgs_per_cu > 1 ? 1024 : 0);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int wgs = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), lds, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double flops = (double)reps * wgs * 4.0 * iters * 9.0 * 16.0 * 2048.0;
    return flops / (ms * 1e-3) / 1e12;
}

extern "C" double mfma_lds_tflops(int mode, int wgs_per_cu, int iters, int reps) {
    switch (mode) {
        case 0: return run<0>(wgs_per_cu, iters, reps);
        case 1: return run<1>(wgs_per_cu, iters, reps);
        case 2: return run<2>(wgs_per_cu, iters, reps);
        case 3: return run<3>(wgs_per_cu, iters, reps);
        default: return run<4>(wgs_per_cu, iters, reps);
    }
}

// ---- mode 5/6: the pure-MFMA ceiling with RANDOM operands (bit toggling costs power; the part clocks to its
// power budget), mode 6 = all-zero operands for contrast
__global__ __launch_bounds__(256) void k_rand(float* out, const float* src, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + i + 2048) & 4095]; }
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

extern "C" double mfma_rand_tflops(int zero, int wgs_per_cu, int iters, int reps) {
    float* out; float* src;
    hipMalloc(&out, 4); hipMalloc(&src, 4096 * 4);
    float h[4096];
    unsigned x = 12345u;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0.f : ((x >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k_rand, dim3(wgs), dim3(256), 0, 0, out, src, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rand, dim3(wgs), dim3(256), 0, 0, out, src, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out); hipFree(src);
    return (double)reps * wgs * 4.0 * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
}
