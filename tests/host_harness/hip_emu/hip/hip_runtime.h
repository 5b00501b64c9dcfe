// TEST INFRASTRUCTURE ONLY: a host-side stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel sources of
// frustum_convnet_amd/csrc compile as plain C++ (clang++ -x c++) and run on the CPU, so the kernels' index arithmetic,
// LDS choreography and reductions can be checked against the oracle without a GPU (tests/test_emu_fcn.py).
// Never part of the product: libfcn_hip.so is built by hipcc for gfx950 only and has no CPU path.
//
// Execution model: workgroups run one after the other; every thread of a workgroup is a ucontext coroutine; a thread runs
// until it reaches a collective (workgroup barrier, wave shuffle, MFMA) and is resumed once all live threads of the
// workgroup / wave have arrived.  `__shared__` becomes `static` (one workgroup at a time).  MFMA builtins are computed
// from the operand registers the 64 lanes deposit, with the gfx950 operand layouts (lane l: row/column l % 32, reduction
// index block l / 32).  Wave-uniform helpers (readfirstlane, SGPR pins) are the identity.
#pragma once
#include <setjmp.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
// `asm volatile("" : "+s"(v));` (an SGPR pin, no code) -> `;`
#define asm
#define volatile(...)
// occupancy hints mean nothing on the host (`__attribute__((amdgpu_waves_per_eu(a, b)))` -> `__attribute__((unused))`)
#define amdgpu_waves_per_eu(...) unused

// dynamic LDS: one buffer per launch, sized by the launch's shared-memory argument
#define FCN_DYN_LDS(T, name) T *name = (T *)emu::dyn_lds

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) x
#define hipMemcpyFromSymbol(dst, sym, n) (memcpy((dst), &(sym), (n)), hipSuccess)
#define hipMemcpyToSymbol(sym, src, n) (memcpy(&(sym), (src), (n)), hipSuccess)

namespace emu {
struct Idx { unsigned x, y, z; };
struct Group { int arrived = 0, gen = 0, alive = 0; };
struct Lane {
    ucontext_t ctx;       // first entry only; later switches go through jb (no signal-mask system call per switch)
    jmp_buf jb;
    Idx tidx;
    int lane, wave;
    bool done, started;
    const Group *wait_g;  // blocked in a rendezvous of this group since generation wait_gen (the scheduler skips it until that can change)
    int wait_gen;
};
constexpr int MAXT = 1024;
constexpr size_t STACK = 512 * 1024;
inline Lane g_lanes[MAXT];
inline char *g_stacks = nullptr;
inline Lane *cur = nullptr;
inline ucontext_t sched;
inline jmp_buf sched_jb;
inline Idx bidx, bdim, gdim;
inline Group wg, waves[MAXT / 64];
inline unsigned char slots[MAXT / 64][64][64];        // per wave, per lane: a collective's deposit (<= 64 bytes)
inline const void *kernarg = nullptr;
inline unsigned char *dyn_lds = nullptr;
inline const std::function<void()> *body = nullptr;
inline long n_switch = 0;

inline void yield() { ++n_switch; if (!_setjmp(cur->jb)) _longjmp(sched_jb, 1); }
inline bool lane_live(int lane) { const int t = cur->wave * 64 + lane; return t < (int)bdim.x && !g_lanes[t].done; }
inline void group_barrier(Group &g)
{
    const int my = g.gen;
    g.arrived++;
    cur->wait_g = &g;
    cur->wait_gen = my;
    while (g.gen == my) {
        if (g.arrived >= g.alive) { g.arrived = 0; g.gen++; break; }
        yield();
    }
    cur->wait_g = nullptr;
}
inline void trampoline()
{
    (*body)();
    cur->done = true;
    wg.alive--;
    waves[cur->wave].alive--;
    _longjmp(sched_jb, 1);
}
template <class F>
inline void launch(dim3 grid, dim3 block, size_t shm, const F &f, const void *karg)
{
    const int nt = (int)(block.x * block.y * block.z);
    if (nt > MAXT || block.y != 1 || block.z != 1) { fprintf(stderr, "emu: unsupported launch shape\n"); abort(); }
    if (!g_stacks) g_stacks = (char *)malloc(STACK * MAXT);
    std::vector<unsigned char> dyn(shm + 64);
    dyn_lds = (unsigned char *)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    const std::function<void()> fn = f;
    body = &fn;
    kernarg = karg;
    bdim = {block.x, 1, 1};
    gdim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned b = 0; b < grid.x; ++b) {
        bidx = {b, by, bz};
        wg = Group();
        wg.alive = nt;
        for (int w = 0; w < (nt + 63) / 64; ++w) { waves[w] = Group(); waves[w].alive = std::min(64, nt - 64 * w); }
        for (int t = 0; t < nt; ++t) {
            Lane &L = g_lanes[t];
            L.tidx = {(unsigned)t, 0, 0};
            L.lane = t & 63;
            L.wave = t >> 6;
            L.done = false;
            L.started = false;
            L.wait_g = nullptr;
            getcontext(&L.ctx);
            L.ctx.uc_stack.ss_sp = g_stacks + STACK * t;
            L.ctx.uc_stack.ss_size = STACK;
            L.ctx.uc_link = &sched;
            makecontext(&L.ctx, (void (*)())trampoline, 0);
        }
        int live = nt;
        while (live > 0) {
            live = 0;
            int ran = 0;
            for (int t = 0; t < nt; ++t) {
                if (g_lanes[t].done) continue;
                cur = &g_lanes[t];
                if (cur->wait_g && cur->wait_g->gen == cur->wait_gen && cur->wait_g->arrived < cur->wait_g->alive) { ++live; continue; }
                ++ran;
                if (!_setjmp(sched_jb)) {
                    if (cur->started) _longjmp(cur->jb, 1);
                    cur->started = true;
                    swapcontext(&sched, &cur->ctx);
                }
                if (!g_lanes[t].done) ++live;
            }
            if (live > 0 && ran == 0) {       // every live thread waits for threads that will never arrive
                fprintf(stderr, "emu: deadlock in workgroup (%u,%u,%u): %d threads blocked in a barrier / wave collective that the "
                                "others never reach (divergent __syncthreads, or a collective under lane-divergent control flow)\n",
                        bidx.x, bidx.y, bidx.z, live);
                abort();
            }
        }
    }
    cur = nullptr;
    body = nullptr;
    dyn_lds = nullptr;
}
template <class T, class... R>
inline const void *first_arg(const T &a, const R &...) { return &a; }

// every live lane of the wave deposits `n` bytes; returns after all have (read others' through peer()); call done() after
inline void deposit(const void *p, size_t n)
{
    memcpy(slots[cur->wave][cur->lane], p, n);
    group_barrier(waves[cur->wave]);
}
inline const void *peer(int lane) { return slots[cur->wave][lane]; }
inline void done() { group_barrier(waves[cur->wave]); }

typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <class H8>
inline f32x16_t mfma_32x32x16(H8 a, H8 b, f32x16_t c)
{
    struct Dep { H8 a, b; } d = {a, b};
    static_assert(sizeof(Dep) <= 64, "deposit slot");
    deposit(&d, sizeof(d));
    const int l31 = cur->lane & 31, lh = cur->lane >> 5;
    // D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[l%32][8*(l/32) + 0..7], B[8*(l/32) + 0..7][l%32]; it owns column
    // j = l % 32 and rows (reg & 3) + 8 * (reg >> 2) + 4 * (l / 32)
    for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        float s = c[reg];
        for (int k = 0; k < 16; ++k) {
            const Dep *pa = (const Dep *)peer(i + 32 * (k >> 3)), *pb = (const Dep *)peer(l31 + 32 * (k >> 3));
            s += (float)pa->a[k & 7] * (float)pb->b[k & 7];
        }
        c[reg] = s;
    }
    done();
    return c;
}
inline f32x16_t mfma_32x32x2_f32(float a, float b, f32x16_t c)
{
    struct Dep { float a, b; } d = {a, b};
    deposit(&d, sizeof(d));
    const int l31 = cur->lane & 31, lh = cur->lane >> 5;
    for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        float s = c[reg];
        for (int k = 0; k < 2; ++k) s = fmaf(((const Dep *)peer(i + 32 * k))->a, ((const Dep *)peer(l31 + 32 * k))->b, s);
        c[reg] = s;
    }
    done();
    return c;
}
template <class T>
inline T shfl_xor(T v, int mask)
{
    static_assert(sizeof(T) <= 64, "deposit slot");
    deposit(&v, sizeof(v));
    T r;
    memcpy(&r, peer((cur->lane ^ mask) & 63), sizeof(T));
    done();
    return r;
}
template <class T>
inline T shfl_from(T v, int src_lane)          // src_lane differs per lane (computed by the caller)
{
    static_assert(sizeof(T) <= 64, "deposit slot");
    deposit(&v, sizeof(v));
    T r;
    memcpy(&r, peer(src_lane & 63), sizeof(T));
    done();
    return r;
}
inline unsigned long long ballot(bool p)
{
    // exited lanes contribute 0 (their slots are cleared when they leave, see trampoline)
    unsigned char b = p ? 1 : 0;
    deposit(&b, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (lane_live(l) && *(const unsigned char *)peer(l)) m |= 1ull << l;
    done();
    return m;
}
inline const char __attribute__((address_space(4))) *kernarg_ptr()
{
    return (const char __attribute__((address_space(4))) *)(uintptr_t)kernarg;
}
}  // namespace emu

#define threadIdx (emu::cur->tidx)
#define blockIdx (emu::bidx)
#define blockDim (emu::bdim)
#define gridDim (emu::gdim)
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...) \
    emu::launch((grid), (block), (size_t)(shm), [&]() { kern(__VA_ARGS__); }, emu::first_arg(__VA_ARGS__))

inline void __syncthreads() { emu::group_barrier(emu::wg); }
inline void __threadfence() {}
inline void __threadfence_block() {}
template <class T>
inline T __shfl_xor(T v, int mask, int = 64) { return emu::shfl_xor(v, mask); }
template <class T>
inline T __shfl_up(T v, unsigned delta, int = 64) { const int l = emu::cur->lane; return emu::shfl_from(v, l >= (int)delta ? l - (int)delta : l); }
template <class T>
inline T __shfl_down(T v, unsigned delta, int = 64) { const int l = emu::cur->lane; return emu::shfl_from(v, l + (int)delta < 64 ? l + (int)delta : l); }
template <class T>
inline T __shfl(T v, int src, int = 64) { return emu::shfl_from(v, src); }
inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline long long clock64() { return 0; }
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// wave-synchronous LDS exchange: on the GPU the 64 lanes execute in lockstep and this builtin only pins the compiler's
// order; here the lanes are coroutines, so it is a real wave-level barrier
#define __builtin_amdgcn_wave_barrier() emu::group_barrier(emu::waves[emu::cur->wave])
#define __builtin_amdgcn_kernarg_segment_ptr() (emu::kernarg_ptr())
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2_f32((a), (b), (c))
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T>
inline T emu_fetch_add(T *p, T v) { const T o = *p; *p = o + v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
inline unsigned atomicAdd(unsigned *p, unsigned v) { return emu_fetch_add(p, v); }
inline int atomicAdd(int *p, int v) { return emu_fetch_add(p, v); }
inline float atomicAdd(float *p, float v) { return emu_fetch_add(p, v); }
inline unsigned long long wall_clock64() { return 0ull; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <class T>
inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
inline T max(T a, T b) { return a < b ? b : a; }
