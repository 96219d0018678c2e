// Which compute units do the workgroups of a CU-masked stream land on?  (dev tool, round 6: VERDICT r05 item 1)
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/cumask_probe.bin tools/micro/cumask_probe.hip && tools/micro/cumask_probe.bin
// For each mask layout the probe launches 4096 long-lived workgroups on a stream made by hipExtStreamCreateWithCUMask,
// every workgroup records (XCC_ID, SE, CU) from the hardware registers, and the host prints the set of distinct
// (xcc, se, cu) triples per XCC — i.e. whether the mask is honoured for an ordinary user on this box and how mask BIT i
// maps to a physical CU (the KFD deals the bits round-robin over the XCCs: bit i -> XCC i % 8).
// A second leg measures what a mask does to a pure-MFMA kernel's time (k CUs -> 256 / k times longer when honoured).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <map>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void where_am_i(uint32_t* out, int spin) {
    // HW_REG_HW_ID (id 4): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se;  HW_REG_XCC_ID (id 20): [3:0] xcc
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a = 1.f + threadIdx.x * 1e-6f;
    for (int i = 0; i < spin; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.f, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16) | (acc[0] == 123.f ? 1u << 31 : 0u);
}

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.f + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.f, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 123.456f) out[0] = s;
}

static std::vector<uint32_t> make_mask(const char* kind, int k) {
    std::vector<uint32_t> m(8, 0u);                     // 256 bits
    for (int i = 0; i < 256; ++i) {
        bool on;
        if (!strcmp(kind, "first")) on = i < k;                         // the first k bits (k / 8 CUs of every XCC if bits are dealt round-robin)
        else if (!strcmp(kind, "xcc")) on = (i % 8) < k / 32;           // k / 32 whole XCCs under the round-robin reading
        else on = true;
        if (on) m[i / 32] |= 1u << (i % 32);
    }
    return m;
}

int main() {
    const int WGS = 4096;
    uint32_t* d; hipMalloc(&d, WGS * 4);
    float* f; hipMalloc(&f, 4);
    std::vector<uint32_t> h(WGS);
    struct Case { const char* kind; int k; } cases[] = {{"all", 256}, {"first", 64}, {"first", 96}, {"first", 128}, {"first", 192},
                                                         {"xcc", 64}, {"xcc", 128}};
    for (auto& c : cases) {
        std::vector<uint32_t> m = make_mask(c.kind, c.k);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
        if (e != hipSuccess) { printf("mask %s/%d: hipExtStreamCreateWithCUMask -> %s\n", c.kind, c.k, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0, WGS * 4, s);
        hipLaunchKernelGGL(where_am_i, dim3(WGS), dim3(256), 0, s, d, 2000);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, WGS * 4, hipMemcpyDeviceToHost);
        std::map<int, std::set<int>> per_xcc;
        for (int i = 0; i < WGS; ++i) {
            const int xcc = (h[i] >> 16) & 0xf, cu = (h[i] >> 8) & 0xf, sh = (h[i] >> 12) & 1, se = (h[i] >> 13) & 7;
            per_xcc[xcc].insert((se << 5) | (sh << 4) | cu);
        }
        int total = 0;
        printf("mask %-5s k=%3d:", c.kind, c.k);
        for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        // pure-MFMA kernel, 2 workgroups per (unmasked) CU: time scales with 256 / (CUs in the mask) when the mask is honoured
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(512), dim3(256), 0, s, f, 20000);
        hipEventRecord(e0, s);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_loop, dim3(512), dim3(256), 0, s, f, 20000);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  -> %d distinct CUs;  512-workgroup MFMA loop %.3f ms per launch\n", total, ms / 3);
        hipStreamDestroy(s);
    }
    // two masked streams side by side: complementary halves should each keep their stand-alone time
    {
        std::vector<uint32_t> m1 = make_mask("first", 128), m2(8);
        for (int i = 0; i < 8; ++i) m2[i] = ~m1[i];
        hipStream_t s1, s2;
        if (hipExtStreamCreateWithCUMask(&s1, 8, m1.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&s2, 8, m2.data()) == hipSuccess) {
            hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
            hipDeviceSynchronize();
            hipEventRecord(a0, s1); hipEventRecord(b0, s2);
            for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, s1, f, 20000);
                                           hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, s2, f, 20000); }
            hipEventRecord(a1, s1); hipEventRecord(b1, s2);
            hipDeviceSynchronize();
            float ma, mb; hipEventElapsedTime(&ma, a0, a1); hipEventElapsedTime(&mb, b0, b1);
            printf("complementary 128 + 128 masks, 256 workgroups each, concurrently: %.3f / %.3f ms per launch\n", ma / 3, mb / 3);
        }
    }
    return 0;
}
