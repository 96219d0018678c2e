// amx_device.h — device prelude shared by every kernel source in atomai_amd/csrc.
//
// Product build:  hipcc --offload-arch=gfx950 (MI355X / CDNA4 only; 64-wide wavefronts).
// Test build:     g++ -DAMX_EMU -include tests/emu/hip_emu.h  (CPU SIMT emulator, `not gpu` tests only).
#pragma once
#ifdef AMX_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define AMX_DYN_SMEM(type, name)                                      \
    extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#define AMX_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#endif
#include <atomic>
#include <cstdint>
#include "knobs.h"

#define AMX_WAVE 64

// Error convention of the C ABI (include/atomai_amd.h): 0 = ok, >0 = hipError_t, <0 = bad argument; the text of
// the last failure on the calling thread is kept for amx_last_error() (abi.hip).
#include <cstdio>
extern "C" void amx_set_error(const char* fn, int code, const char* detail);
#define AMX_BADARG(code)                                          \
    do {                                                          \
        amx_set_error(__func__, -(code), "bad argument group");   \
        return -(code);                                           \
    } while (0)
#ifdef AMX_EMU
#define AMX_ERRSTR(e) "launch failed (emulator)"
#else
#define AMX_ERRSTR(e) hipGetErrorString(e)
#endif
#define AMX_CHECK_LAUNCH()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) {                                  \
            amx_set_error(__func__, (int)e__, AMX_ERRSTR(e__));   \
            return (int)e__;                                      \
        }                                                         \
    } while (0)

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised — once per
// kernel AND device (a process may drive several: SegPredictor(device='cuda:1'), DKL replicas), from whichever host
// thread launches first: a per-call-site atomic bit mask over the device index, no lock.
#ifdef AMX_EMU
#define AMX_ALLOW_160K_LDS(...) do { } while (0)
#else
#define AMX_ALLOW_160K_LDS(...)                                                                          \
    do {                                                                                                 \
        static std::atomic<unsigned> done__{0};                                                          \
        int dev__ = 0;                                                                                   \
        (void)hipGetDevice(&dev__);                                                                      \
        const unsigned bit__ = 1u << (dev__ & 31);                                                       \
        if (!(done__.load(std::memory_order_acquire) & bit__)) {                                         \
            hipError_t e__ = hipFuncSetAttribute((const void*)(__VA_ARGS__),                             \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e__ != hipSuccess) {                                                                     \
                amx_set_error(__func__, (int)e__, AMX_ERRSTR(e__));                                      \
                return (int)e__;                                                                         \
            }                                                                                            \
            done__.fetch_or(bit__, std::memory_order_release);                                           \
        }                                                                                                \
    } while (0)
#endif

// Wave-level ordering point between LDS writes and reads of OTHER lanes of the same wave.  The hardware executes a
// wave's instructions in lock step, so only the compiler must be kept from reordering (wave_barrier emits no code);
// the CPU emulator runs one lane at a time and needs a real rendezvous.
static __device__ __forceinline__ void amx_wave_sync() {
#ifdef AMX_EMU
    (void)__shfl_xor(0.f, 1);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
static __device__ __forceinline__ float4 amx_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static __device__ __forceinline__ void amx_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// XCD-aware block index: the dispatcher places block b on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup
// dispatch"); this bijective remap gives every XCD one CONTIGUOUS range of logical blocks instead, so that blocks
// which share halo rows / columns of their input hit the same 4 MiB L2.  Speed only — never correctness.
static __device__ __forceinline__ unsigned amx_xcd_remap(unsigned b, unsigned nb) {
    const unsigned xcd = b & 7u, idx = b >> 3, q = nb >> 3, r = nb & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

// Streaming (non-temporal) 16-byte store for outputs that are written once and not re-read by the same kernel
static __device__ __forceinline__ void amx_st4_stream(float* p, float4 v) {
#ifdef AMX_EMU
    *reinterpret_cast<float4*>(p) = v;
#else
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
#endif
}
// Compute units of the current device (256 on MI355X), queried once per device.
static inline int amx_num_cus() {
#ifdef AMX_EMU
    return 4;
#else
    static std::atomic<int> cached[16];              // zero-initialised; racing first callers store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
#endif
}
static __host__ __device__ __forceinline__ int amx_ceil_div(int a, int b) { return (a + b - 1) / b; }
static __host__ __device__ __forceinline__ int amx_round_up(int a, int b) { return (a + b - 1) / b * b; }

// Dilations 2 / 4 / 6 run as d*d plain 3x3 convolutions on the residue-class sub-images (conv_kernel.h, wgrad_kernel.h);
// AMX_CONV_LATTICE=0 selects the halo-class kernels instead (defined in conv_fwd.hip).
bool amx_lattice_mode(int taps, int dil);

