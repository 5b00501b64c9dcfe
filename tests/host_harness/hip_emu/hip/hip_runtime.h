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
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local      // one workgroup per host thread at a time
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
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
inline hipError_t hipStreamGetCaptureInfo(hipStream_t, hipStreamCaptureStatus *st, unsigned long long *id)
{
    *st = hipStreamCaptureStatusNone; *id = 0; return hipSuccess;          // (the emulation runs every launch at once: no captures)
}
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
// per OS thread (a worker runs one workgroup at a time): the workgroup's coroutines and rendezvous state
inline thread_local Lane *g_lanes = nullptr;
inline thread_local char *g_stacks = nullptr;
inline thread_local Lane *cur = nullptr;
inline thread_local ucontext_t sched;
inline thread_local jmp_buf sched_jb;
inline thread_local Idx bidx;
inline thread_local Group wg, waves[MAXT / 64];
inline thread_local unsigned char (*slots)[64][64] = nullptr;   // per wave, per lane: a collective's deposit (<= 64 bytes)
inline thread_local unsigned char *dyn_lds = nullptr;
inline thread_local std::vector<unsigned char> *dyn_buf = nullptr;
// per launch, shared by the workers (read-only while it runs)
inline Idx bdim, gdim;
inline const void *kernarg = nullptr;
inline const std::function<void()> *body = nullptr;
inline size_t job_shm = 0;
inline int job_nt = 0;
inline std::atomic<long> job_next{0};
inline long job_total = 0;

inline void yield() { if (!_setjmp(cur->jb)) _longjmp(sched_jb, 1); }
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
// one workgroup, start to end, on the calling OS thread
inline void run_workgroup(long id)
{
    const int nt = job_nt;
    if (!g_stacks) {
        g_stacks = (char *)malloc(STACK * MAXT);
        g_lanes = new Lane[MAXT];
        slots = (unsigned char (*)[64][64])malloc(sizeof(unsigned char) * (MAXT / 64) * 64 * 64);
        dyn_buf = new std::vector<unsigned char>();
    }
    if (dyn_buf->size() < job_shm + 64) dyn_buf->resize(job_shm + 64);
    dyn_lds = (unsigned char *)(((uintptr_t)dyn_buf->data() + 63) & ~(uintptr_t)63);
    const long gxy = (long)gdim.x * gdim.y;
    bidx = {(unsigned)(id % gdim.x), (unsigned)((id % gxy) / gdim.x), (unsigned)(id / gxy)};
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
    cur = nullptr;
}
inline void drain_job()
{
    for (;;) {
        const long id = job_next.fetch_add(1);
        if (id >= job_total) break;
        run_workgroup(id);
    }
}
// Persistent workers (FCN_EMU_THREADS, default min(8, cores)): the workgroups of a launch are independent except for atomics
// (real ones here) and "last workgroup" tickets, so they run on several host cores; a launch returns when all are done.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long gen = 0;
    int busy = 0;
    bool stop = false;
    int n = 1;
    Pool()
    {
        const char *e = getenv("FCN_EMU_THREADS");
        n = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        if (n < 1) n = 1;
        for (int i = 1; i < n; ++i)
            th.emplace_back([this] {
                long seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                    }
                    drain_job();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--busy == 0) cv_done.notify_all();
                    }
                }
            });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    void run()
    {
        if (n == 1 || job_total < 4) { drain_job(); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            busy = n - 1;
            ++gen;
        }
        cv_go.notify_all();
        drain_job();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return busy == 0; });
    }
};
inline Pool &pool() { static Pool p; return p; }

template <class F>
inline void launch(dim3 grid, dim3 block, size_t shm, const F &f, const void *karg)
{
    const int nt = (int)(block.x * block.y * block.z);
    if (nt > MAXT || block.y != 1 || block.z != 1) { fprintf(stderr, "emu: unsupported launch shape\n"); abort(); }
    const std::function<void()> fn = f;
    body = &fn;
    kernarg = karg;
    bdim = {block.x, 1, 1};
    gdim = {grid.x, grid.y, grid.z};
    job_shm = shm;
    job_nt = nt;
    job_total = (long)grid.x * grid.y * grid.z;
    job_next.store(0);
    pool().run();
    body = nullptr;
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
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
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
template <class T, class V>
inline void emu_atomic_store(T *p, V v) { T t = (T)v; __atomic_store(p, &t, __ATOMIC_SEQ_CST); }
template <class T>
inline T emu_atomic_load(const T *p) { T t; __atomic_load(const_cast<T *>(p), &t, __ATOMIC_SEQ_CST); return t; }
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v))
#define __hip_atomic_load(p, order, scope) emu_atomic_load((p))
#define FCN_HOST_EMU 1                                  // the kernel sources compile their C++ restatement of inline-asm helpers
// v_med3_f32 (IEEE mode): a NaN operand makes it min3 over the others' ordering (fminf drops the NaN)
inline float emu_fmed3f(float a, float b, float c)
{
    if (a != a || b != b || c != c) return fminf(fminf(a, b), c);
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3f((a), (b), (c))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))           // v_rcp_f32 (1 ulp on the GPU; only used where a correction step follows)
// wave-synchronous LDS exchange: on the GPU the 64 lanes execute in lockstep and this builtin only pins the compiler's
// order; here the lanes are coroutines, so it is a real wave-level barrier
#define __builtin_amdgcn_wave_barrier() emu::group_barrier(emu::waves[emu::cur->wave])
#define __builtin_amdgcn_kernarg_segment_ptr() (emu::kernarg_ptr())
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2_f32((a), (b), (c))
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T>
inline T emu_fetch_add(T *p, T v)          // real atomics: workgroups run on several host threads
{
    if constexpr (std::is_integral<T>::value) {
        return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
    } else {
        typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
        U *q = (U *)p;
        U o = __atomic_load_n(q, __ATOMIC_SEQ_CST);
        for (;;) {
            T ov;
            memcpy(&ov, &o, sizeof(T));
            const T nv = ov + v;
            U n;
            memcpy(&n, &nv, sizeof(T));
            if (__atomic_compare_exchange_n(q, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return ov;
        }
    }
}
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
inline unsigned atomicAdd(unsigned *p, unsigned v) { return emu_fetch_add(p, v); }
inline int atomicAdd(int *p, int v) { return emu_fetch_add(p, v); }
inline float atomicAdd(float *p, float v) { return emu_fetch_add(p, v); }
inline int atomicOr(int *p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v)
{
    unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
#define __HIP_MEMORY_SCOPE_WORKGROUP 1
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned long long wall_clock64() { return 0ull; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <class T>
inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
inline T max(T a, T b) { return a < b ? b : a; }
