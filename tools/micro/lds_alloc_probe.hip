// What does HW_REG_LDS_ALLOC say for two workgroups that share a CU?  (dev tool, round 6: the rDecoder forward kernel's
// RD_FWD_ASYM experiment keys on a non-zero LDS base.)   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_alloc_probe.bin tools/micro/lds_alloc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void probe(uint32_t* out, int spin) {
    extern __shared__ float s[];
    uint32_t la, hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s[threadIdx.x] = threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < spin; ++i) { __syncthreads(); acc += s[(threadIdx.x + i) & 511]; }
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = la; out[blockIdx.x * 3 + 1] = (hw & 0xffff) | ((xcc & 0xf) << 16); out[blockIdx.x * 3 + 2] = acc == 1.5f; }
}
int main() {
    const int WGS = 512;
    uint32_t* d; hipMalloc(&d, WGS * 12);
    const size_t lds = 78 * 1024;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe, dim3(WGS), dim3(512), lds, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(WGS * 3);
    hipMemcpy(h.data(), d, WGS * 12, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> hist;
    std::map<uint32_t, std::vector<int>> per_cu;
    for (int i = 0; i < WGS; ++i) { hist[h[i * 3]]++; per_cu[h[i * 3 + 1] & 0xfff00].push_back(i); }
    for (auto& kv : hist) printf("LDS_ALLOC 0x%08x  (low 12 bits 0x%03x, bits 12.. 0x%x): %d workgroups\n", kv.first, kv.first & 0xfff, kv.first >> 12, kv.second);
    int shown = 0;
    for (auto& kv : per_cu) { if (shown++ < 6) { printf("CU key 0x%05x: workgroups", kv.first); for (int b : kv.second) printf(" %d(la %03x)", b, h[b * 3] & 0xfff); printf("\n"); } }
    printf("%zu distinct CUs host the %d workgroups\n", per_cu.size(), WGS);
    return 0;
}
