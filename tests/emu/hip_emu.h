// hip_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal single-threaded SIMT emulator that lets the *same* kernel sources under
// atomai_amd/csrc/*.hip be compiled with g++ (-DAMX_EMU) and executed on the CPU for the
// `not gpu` test tier (there is no GPU in the dev container).  Every HIP thread of a block is a
// fiber (ucontext); __syncthreads() and the wave-level collectives (shuffles, MFMA) are
// rendezvous points between fibers.  MFMA is emulated as the k-ordered fmaf chain the gfx950
// hardware produces (cdna_hip_programming.md §3 "Numerics"), with the documented lane<->element
// maps, so index/layout bugs show up here exactly as they would on the GPU.
//
// Nothing in the product package loads the library built from this header; the tests inject it
// explicitly (tests/emu/emu_backend.py).
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

struct Wave {
    int count = 0;
    unsigned gen = 0;
    unsigned seq = 0;                 // parity selects the exchange slot
    float a[2][WAVE], b[2][WAVE];
    uint64_t bits[2];
};

// Fiber context.  On x86-64 a 20-instruction user-space switch (callee-saved registers + MXCSR/x87 CW);
// glibc's swapcontext makes an rt_sigprocmask system call per switch, which dominated the emulator's run time
// (an emulated MFMA is a 64-fiber rendezvous).  Other hosts keep ucontext.
#if defined(__x86_64__)
struct Ctx { void* sp = nullptr; };
__attribute__((naked, noinline)) static void ctx_switch(Ctx* /*from: rdi*/, Ctx* /*to: rsi*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "subq $8, %rsp\n\tstmxcsr (%rsp)\n\tfnstcw 4(%rsp)\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq (%rsi), %rsp\n\t"
        "ldmxcsr (%rsp)\n\tfldcw 4(%rsp)\n\taddq $8, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}
static inline void ctx_make(Ctx& c, char* stack, size_t bytes, void (*entry)()) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(stack) + bytes) & ~uintptr_t(15);
    uint64_t* sp = reinterpret_cast<uint64_t*>(top - 16);     // return address slot: entry sees rsp % 16 == 8
    *sp = reinterpret_cast<uint64_t>(entry);
    for (int i = 0; i < 6; ++i) *--sp = 0;                    // rbp rbx r12 r13 r14 r15
    --sp;
    *reinterpret_cast<uint32_t*>(sp) = 0x1F80u;               // MXCSR default
    *(reinterpret_cast<uint16_t*>(sp) + 2) = 0x037Fu;         // x87 control word default
    *(reinterpret_cast<uint16_t*>(sp) + 3) = 0;
    c.sp = sp;
}
#else
struct Ctx { ucontext_t uc; };
static inline void ctx_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
static inline void ctx_make(Ctx& c, char* stack, size_t bytes, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack; c.uc.uc_stack.ss_size = bytes; c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
    Ctx ctx;
    dim3 tid;
    int linear = 0;
    int lane = 0;
    int wave = 0;
    bool done = false;
    unsigned my_seq = 0;              // per-lane collective counter (must match wave.seq order)
};

struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<Wave> waves;
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    std::vector<char> dyn_smem;
    Ctx sched;
    Fiber* cur = nullptr;
    std::function<void()> body;
};

inline Block& blk() { static Block b; return b; }

inline void yield() { Block& B = blk(); ctx_switch(&B.cur->ctx, &B.sched); }

inline void trampoline() {
    Block& B = blk();
    B.body();
    B.cur->done = true;
    ctx_switch(&B.cur->ctx, &B.sched);
    abort();                                                  // a finished fiber is never resumed
}

template <class F>
void launch(dim3 grid, dim3 block, size_t shmem, F&& f) {
    Block& B = blk();
    B.gdim = grid; B.bdim = block;
    B.nthreads = block.x * block.y * block.z;
    if (B.nthreads % WAVE != 0) { fprintf(stderr, "emu: block size must be a multiple of 64\n"); abort(); }
    B.body = f;
    B.dyn_smem.assign(shmem + 64, 0);
    if (B.stacks.size() < (size_t)B.nthreads * STACK_BYTES) B.stacks.resize((size_t)B.nthreads * STACK_BYTES);
    B.fibers.resize(B.nthreads);
    B.waves.assign(B.nthreads / WAVE, Wave());
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        B.bid = dim3(bx, by, bz);
        B.bar_count = 0; B.bar_gen = 0;
        for (auto& w : B.waves) w = Wave();
        for (int t = 0; t < B.nthreads; ++t) {
            Fiber& fb = B.fibers[t];
            fb.linear = t; fb.lane = t % WAVE; fb.wave = t / WAVE; fb.done = false; fb.my_seq = 0;
            fb.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            ctx_make(fb.ctx, B.stacks.data() + (size_t)t * STACK_BYTES, STACK_BYTES, trampoline);
        }
        int remaining = B.nthreads;
        while (remaining > 0) {
            for (int t = 0; t < B.nthreads; ++t) {
                Fiber& fb = B.fibers[t];
                if (fb.done) continue;
                B.cur = &fb;
                ctx_switch(&B.sched, &fb.ctx);
                if (fb.done) --remaining;
            }
        }
    }
}

inline void syncthreads() {
    Block& B = blk();
    unsigned gen = B.bar_gen;
    if (++B.bar_count == B.nthreads) { B.bar_count = 0; ++B.bar_gen; }
    else while (B.bar_gen == gen) yield();
}

// Rendezvous of the 64 lanes of the calling wave.  All lanes must be active (the kernels in this
// repo only use wave collectives in wave-uniform control flow).
inline void wave_sync() {
    Block& B = blk();
    Wave& W = B.waves[B.cur->wave];
    unsigned gen = W.gen;
    if (++W.count == WAVE) { W.count = 0; ++W.gen; }
    else while (W.gen == gen) yield();
}

inline float shfl_idx(float v, int src_lane) {
    Block& B = blk(); Fiber* me = B.cur; Wave& W = B.waves[me->wave];
    int s = me->my_seq++ & 1;
    W.a[s][me->lane] = v;
    wave_sync();
    return W.a[s][src_lane & (WAVE - 1)];
}

inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    Block& B = blk(); Fiber* me = B.cur; Wave& W = B.waves[me->wave];
    int s = me->my_seq++ & 1;
    W.a[s][me->lane] = a; W.b[s][me->lane] = b;
    wave_sync();
    const int j = me->lane & 15, g = me->lane >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        const int row = 4 * g + r;                       // C/D: col = lane&15, row = 4*(lane>>4)+reg
        for (int k = 0; k < 4; ++k)                      // A[i][k] in lane i+16k, B[k][j] in lane j+16k
            acc = fmaf(W.a[s][row + 16 * k], W.b[s][j + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks, K = 1.  Lane map measured on the MI355X
// (tools/micro/mfma4x4_probe.hip): D[lane l][reg r] = A[lane 4*(l/4) + r] * B[lane l] + C[l][r]
// (block = l/4, row = reg, column = l%4; A's row index is its lane%4, B's column index its lane%4).
inline f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
    Block& B = blk(); Fiber* me = B.cur; Wave& W = B.waves[me->wave];
    int s = me->my_seq++ & 1;
    W.a[s][me->lane] = a; W.b[s][me->lane] = b;
    wave_sync();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) d[r] = fmaf(W.a[s][4 * (me->lane >> 2) + r], W.b[s][me->lane], c[r]);
    return d;
}

inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    Block& B = blk(); Fiber* me = B.cur; Wave& W = B.waves[me->wave];
    int s = me->my_seq++ & 1;
    W.a[s][me->lane] = a; W.b[s][me->lane] = b;
    wave_sync();
    const int j = me->lane & 31, h = me->lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // C/D: col = lane&31, row=(reg&3)+8*(reg>>2)+4*(lane>>5)
        float acc = c[r];
        for (int k = 0; k < 2; ++k)                      // A[i][k] in lane i+32k, B[k][j] in lane j+32k
            acc = fmaf(W.a[s][row + 32 * k], W.b[s][j + 32 * k], acc);
        d[r] = acc;
    }
    return d;
}

}  // namespace emu

#define threadIdx (emu::blk().cur->tid)
#define blockIdx (emu::blk().bid)
#define blockDim (emu::blk().bdim)
#define gridDim (emu::blk().gdim)
#define warpSize 64

static inline void __syncthreads() { emu::syncthreads(); }
static inline float __shfl_xor(float v, int mask) { return emu::shfl_idx(v, emu::blk().cur->lane ^ mask); }
static inline double __shfl_xor(double v, int mask) {      // (two 32-bit halves through the float exchange: bit patterns are copied, never computed on)
    float h[2]; memcpy(h, &v, 8);
    const int src = emu::blk().cur->lane ^ mask;
    h[0] = emu::shfl_idx(h[0], src); h[1] = emu::shfl_idx(h[1], src);
    double o; memcpy(&o, h, 8); return o;
}
static inline float __shfl_down(float v, int d) { int l = emu::blk().cur->lane; return emu::shfl_idx(v, l + d < 64 ? l + d : l); }
static inline float __shfl(float v, int src) { return emu::shfl_idx(v, src); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) { return emu::mfma_16x16x4(a, b, c); }
static inline f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4 c, int, int, int) { return emu::mfma_4x4x1(a, b, c); }
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) { return emu::mfma_32x32x2(a, b, c); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline void __threadfence() {}

#define AMX_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>((reinterpret_cast<uintptr_t>(emu::blk().dyn_smem.data()) + 63) & ~uintptr_t(63))
#define AMX_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); })
#define AMX_LAUNCH_T(kernel, grid, block, shmem, stream, ...) AMX_LAUNCH(kernel, grid, block, shmem, stream, __VA_ARGS__)
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
