// Lane <-> element map of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1), found by experiment:
//   run 1: a[lane] = lane, b = 1  -> D tells which A lane feeds each (lane, reg)
//   run 2: a = 1, b[lane] = lane  -> which B lane
// hipcc --offload-arch=gfx950 -o tools/micro/mfma4x4_probe.bin tools/micro/mfma4x4_probe.hip && tools/micro/mfma4x4_probe.bin (result: D[lane l][reg r] = A[lane 4*(l/4)+r] * B[lane l])
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
    const int lane = threadIdx.x;
    const float a = mode == 0 ? (float)lane : 1.f, b = mode == 0 ? 1.f : (float)lane;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    float h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s lane feeding D[lane][reg]):\n", mode, mode == 0 ? "A" : "B");
        for (int lane = 0; lane < 64; ++lane)
            printf("  lane %2d: %2.0f %2.0f %2.0f %2.0f%s", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3], lane % 4 == 3 ? "\n" : " |");
    }
    return 0;
}
