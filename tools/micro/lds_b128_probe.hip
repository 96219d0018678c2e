// LDS bank-conflict probe (round 6): the conv kernels' LDS address patterns in isolation, one kernel
// per pattern, so that a PMC pass (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_LDS_ADDR_CONFLICT per kernel name) says
// which of them conflict and by how much.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_b128_probe.bin tools/micro/lds_b128_probe.hip
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/lds_probe -- tools/micro/lds_b128_probe.bin
//   (tools/gpu_pmc_lds.sh runs both counter passes and summarises them into profiles/r06_lds_conflicts.md)
// Patterns (every kernel: 256 threads = 4 waves, lane -> (p = lane & 15, g = lane >> 4), ITER repetitions, results kept
// alive through a never-true store):
//   frag_read<PLANE>        conv_kernel.h / conv_ws.hip A-fragment read: ds_read_b128 at ((g * PLANE + row * IW + p) * 16 B)
//                           PLANE = 192 (conv_kernel.h 8-row tile: == 0 mod 16), 336 (== 0 mod 16), 338 / 340 (conv_ws.hip)
//   wfrag_read<NB>          B-fragment read: ((tap * 4 + g) * NB + p) * 16 B, NB = 32 (== 0 mod 16) and 28 (remainder class)
//   stage_write<PLANE, MAP> staging ds_write_b128: MAP 0 = channel group fastest (tid & 3 = group, tid >> 2 = pixel slot:
//                           conv_kernel.h), MAP 1 = 8 consecutive slots per 8 lanes (candidate), PLANE as above
//   ws_stage_write<G,PLANE> conv_ws.hip producers: c8 = tid % G, slot = tid / G
//   epi_store<NBE>          epilogue transpose, scalar ds_write_b32 at ((4 g + r) * NBE + q * 16 + p) * 4 B; NBE = 32, 36
//   epi_read<NBE>           its b128 read-back: consecutive lanes, consecutive 16-B groups
//   hand_store<PS>          conv_ws.hip hand-over: ds_write_b128 at ((q * 16 + p) * PS + row * 16 + 4 g) * 4 B (product: PS = 260)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PLANE>
__global__ __launch_bounds__(256) void frag_read(float* out) {
    extern __shared__ float4 s4[];
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * PLANE; i += 256) s4[i] = make_float4(i, 0, 0, 0);
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int slot = (wave * 2 + tap / 3) * 18 + p + tap % 3;
            const float4 v = s4[g * PLANE + slot];
            acc += v.x;
            asm volatile("" :: "v"(acc));
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int NB>
__global__ __launch_bounds__(256) void wfrag_read(float* out) {
    extern __shared__ float4 s4[];
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 36 * NB; i += 256) s4[i] = make_float4(i, 0, 0, 0);
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4 v = s4[(tap * 4 + g) * NB + p];
            acc += v.x;
            asm volatile("" :: "v"(acc));
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int PLANE, int MAP>
__global__ __launch_bounds__(256) void stage_write(float* out) {
    extern __shared__ float4 s4[];
    const int tid = threadIdx.x;
    // 180 slots of a (8 + 2) x 18 tile x 4 channel groups = 720 float4 = 3 stores per thread (the last partly idle)
    int kg[3], pix[3];
    for (int i = 0; i < 3; ++i) {
        const int e = tid + i * 256;
        if (MAP == 0) { kg[i] = e & 3; pix[i] = e >> 2; }
        else { kg[i] = (e >> 3) & 3; pix[i] = (e & 7) + 8 * (e >> 5); }
    }
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < 3; ++i)
            if (pix[i] < 180) s4[kg[i] * PLANE + pix[i]] = make_float4(it, tid, i, 0);
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s4[tid].x == 1.2345f) out[0] = s4[tid].y;
}

template <int G, int PLANE, int MAP>
__global__ __launch_bounds__(512) void ws_stage_write(float* out) {
    extern __shared__ float4 s4[];
    const int tid = threadIdx.x;
    constexpr int XLD = (324 * G + 511) / 512;
    int c8, slot0, stride;
    if (MAP == 0) { c8 = tid % G; slot0 = tid / G; stride = 512 / G; }
    else { c8 = (tid >> 3) % G; slot0 = (tid & 7) + 8 * (tid / (8 * G)); stride = 512 / G; }
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < XLD; ++i) {
            const int slot = slot0 + i * stride;
            if (slot < 324) s4[c8 * PLANE + slot] = make_float4(it, tid, i, 0);
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s4[tid].x == 1.2345f) out[0] = s4[tid].y;
}

template <int NBE>
__global__ __launch_bounds__(256) void epi_store(float* out) {
    extern __shared__ float s1[];
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    float* s_epi = s1 + wave * (2 * 16 * NBE);
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int mm = 0; mm < 2; ++mm)
            #pragma unroll
            for (int q = 0; q < 2; ++q)
                #pragma unroll
                for (int r = 0; r < 4; ++r) s_epi[(mm * 16 + 4 * g + r) * NBE + q * 16 + p] = (float)(it + r);
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s1[threadIdx.x] == 1.2345f) out[0] = 1.f;
}

template <int NBE>
__global__ __launch_bounds__(256) void epi_read(float* out) {
    extern __shared__ float s1[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 2 * 16 * NBE; i += 256) s1[i] = (float)i;
    __syncthreads();
    const float* s_epi = s1 + wave * (2 * 16 * NBE);
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = k * 64 + lane, pix = e / 8, cgp = e - pix * 8;        // 32 couts = 8 float4 groups per pixel
            const float4 v = *reinterpret_cast<const float4*>(s_epi + pix * NBE + cgp * 4);
            acc += v.x;
            asm volatile("" :: "v"(acc));
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int PS>
__global__ __launch_bounds__(512) void hand_store(float* out) {
    extern __shared__ float s1[];
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int m = 0; m < 2; ++m)
            #pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<float4*>(s1 + (q * 16 + p) * PS + (wave * 2 + m) * 16 + 4 * g) = make_float4(it, m, q, 0);
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s1[threadIdx.x] == 1.2345f) out[0] = 1.f;
}

#define RUN(kernel, threads, lds)                                                               \
    do {                                                                                        \
        CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), lds, 0, d);                         \
        CHECK(hipDeviceSynchronize());                                                          \
        printf("ran %s\n", #kernel);                                                            \
    } while (0)

int main() {
    float* d; CHECK(hipMalloc(&d, 64));
    RUN((frag_read<192>), 256, 4 * 192 * 16);
    RUN((frag_read<336>), 256, 4 * 344 * 16);
    RUN((frag_read<338>), 256, 4 * 344 * 16);
    RUN((frag_read<340>), 256, 4 * 344 * 16);
    RUN((wfrag_read<32>), 256, 36 * 32 * 16);
    RUN((wfrag_read<28>), 256, 36 * 32 * 16);
    RUN((stage_write<192, 0>), 256, 4 * 192 * 16);
    RUN((stage_write<192, 1>), 256, 4 * 192 * 16);
    RUN((ws_stage_write<4, 340, 0>), 512, 8 * 344 * 16);
    RUN((ws_stage_write<8, 338, 0>), 512, 8 * 344 * 16);
    RUN((ws_stage_write<4, 336, 1>), 512, 8 * 344 * 16);
    RUN((ws_stage_write<8, 336, 1>), 512, 8 * 344 * 16);
    RUN((epi_store<32>), 256, 4 * 2 * 16 * 36 * 4);
    RUN((epi_store<36>), 256, 4 * 2 * 16 * 36 * 4);
    RUN((epi_read<32>), 256, 4 * 2 * 16 * 36 * 4);
    RUN((epi_read<36>), 256, 4 * 2 * 16 * 36 * 4);
    RUN((hand_store<260>), 512, 32 * 288 * 4);
    RUN((hand_store<264>), 512, 32 * 288 * 4);
    RUN((hand_store<268>), 512, 32 * 288 * 4);
    RUN((hand_store<272>), 512, 32 * 288 * 4);
    RUN((hand_store<276>), 512, 32 * 288 * 4);
    RUN((hand_store<288>), 512, 32 * 288 * 4);
    return 0;
}
