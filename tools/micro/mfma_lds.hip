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

// ---- mode 7/8: WAVE-SPECIALISED workgroup — the experiment behind a producer/consumer convolution kernel.
// Waves [0, NC) are consumers: per phase 2 chunks x 9 bursts of (4 ds_read_b128 + 16 MFMA) = 288 MFMAs, i.e. the matrix
// work of one 8x16-pixel tile of a 32->32 3x3 layer for one wave (MTW = NT = 2).  Waves [NC, 2 NC) are producers: per
// phase they fetch `NLD` float4 per lane from a big global array (the next tile), run `NVALU` dependent-free FMAs on them
// (index arithmetic + BatchNorm affine + activation + statistics of the real kernel), write them to LDS, read 4 float4
// back (the accumulators handed over by the consumers) and store 4 float4 to global (the finished tile).  One
// __syncthreads() per phase.  The question: how close to the pure-MFMA ceiling does the pipe stay?
template <int NC, int NLD, int NVALU>
__global__ __launch_bounds__(NC * 128) void k_spec(float* out, const float4* src, float4* dst, int iters, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += NC * 128) smem[i] = src[i & 1023].x;
    __syncthreads();
    if (wave < NC) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4* base = reinterpret_cast<const float4*>(smem) + lane;
        float4 f[4];
        for (int it = 0; it < iters; ++it) {
            #pragma unroll
            for (int t = 0; t < 18; ++t) {
                #pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = base[((t * 4 + j) & 7) * 64 + ((it & 1) ? 512 : 0)];
                #define M(C) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0].C, f[2].C, acc[0], 0, 0, 0); \
                             acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[0].C, f[3].C, acc[1], 0, 0, 0); \
                             acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1].C, f[2].C, acc[2], 0, 0, 0); \
                             acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[1].C, f[3].C, acc[3], 0, 0, 0);
                M(x) M(y) M(z) M(w)
                #undef M
            }
            // hand the accumulators over (4 ds_write_b128 per lane) and start the next tile from zero
            float4* hand = reinterpret_cast<float4*>(smem + 4096) + (wave * 64 + lane) * 4;
            #pragma unroll
            for (int i = 0; i < 4; ++i) { hand[i] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]); acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            __syncthreads();
        }
    } else {
        const int pw = wave - NC;
        float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
        size_t tile = (size_t)blockIdx.x * 977u;
        for (int it = 0; it < iters; ++it) {
            float4 r[NLD];
            tile = (tile + 7919u) % (size_t)ntiles;
            #pragma unroll
            for (int i = 0; i < NLD; ++i) r[i] = src[tile * 2048 + (size_t)((i * NC * 64 + pw * 64 + lane) & 2047)];
            float4 h[4];
            const float4* hand = reinterpret_cast<const float4*>(smem + 4096) + (pw * 64 + lane) * 4;
            #pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = hand[i];
            // VALU ballast on the handed-over tile (bias, activation, statistics, transposition of the real epilogue)
            #pragma unroll
            for (int v = 0; v < NVALU / 2; ++v) { float* q = reinterpret_cast<float*>(h); q[v & 15] = __builtin_fmaf(q[v & 15], 1.0001f, 0.5f); }   // 16 independent chains
            #pragma unroll
            for (int i = 0; i < 4; ++i) dst[tile * 1024 + (size_t)((i * NC * 64 + pw * 64 + lane) & 1023)] = h[i];
            // VALU ballast on the fetched tile (affine + activation), then stage it
            #pragma unroll
            for (int v = 0; v < NVALU / 2; ++v) { float* q = reinterpret_cast<float*>(r); q[v % (4 * NLD)] = __builtin_fmaf(q[v % (4 * NLD)], 1.0001f, 0.5f); }
            float4* stage = reinterpret_cast<float4*>(smem + ((it & 1) ? 0 : 2048));
            #pragma unroll
            for (int i = 0; i < NLD; ++i) stage[(i * NC * 64 + pw * 64 + lane) & 511] = r[i];
            keep.x += r[0].x;
            __syncthreads();
        }
        if (keep.x == 123.456f) out[0] = keep.x;
    }
}

template <int NC, int NLD, int NVALU>
static double run_spec(int wgs_per_cu, int iters, int reps) {
    float* out; float4* src; float4* dst;
    const int ntiles = 16384;
    hipMalloc(&out, 4); hipMalloc(&src, (size_t)ntiles * 2048 * 16); hipMalloc(&dst, (size_t)ntiles * 1024 * 16);
    hipMemset(src, 0x3c, (size_t)ntiles * 2048 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) / 1024 * 1024 - (wgs_per_cu > 1 ? 1024 : 0);
    auto kern = k_spec<NC, NLD, NVALU>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int wgs = 256 * wgs_per_cu;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NC * 128), lds, 0, out, src, dst, iters, ntiles);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NC * 128), lds, 0, out, src, dst, iters, ntiles);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out); hipFree(src); hipFree(dst);
    return (double)reps * wgs * NC * iters * 288.0 * 2048.0 / (ms * 1e-3) / 1e12;
}

// variant 0: 4+4 waves, 6 loads, 600 VALU; 1: 8+8 waves (one workgroup per CU); 2: 4+4, 1200 VALU; 3: 4+4, 6 loads, 0 VALU
extern "C" double mfma_spec_tflops(int variant, int wgs_per_cu, int iters, int reps) {
    switch (variant) {
        case 0: return run_spec<4, 6, 600>(wgs_per_cu, iters, reps);
        case 1: return run_spec<8, 6, 600>(wgs_per_cu, iters, reps);
        case 2: return run_spec<4, 6, 1200>(wgs_per_cu, iters, reps);
        default: return run_spec<4, 6, 0>(wgs_per_cu, iters, reps);
    }
}
