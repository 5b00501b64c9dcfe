// Micro-benchmark (tools only): how does MI355X / ROCm schedule a latency-bound CHAIN of small dependent launches (the
// ConvFeatNet forward / backward: 12-13 launches of 560-2000 workgroups) beside a FLOOD of machine-filling launches (the PointNet
// GEMMs: 4480 workgroups each)?  Scenarios: each alone; both eager on two streams (normal / high priority chain stream, CU-masked
// streams); both as hipGraphs launched on those streams; one two-branch graph (default / node priorities).
// Prints the chain's duration and the whole pair's duration per scenario.
//   hipcc --offload-arch=gfx950 -O3 sched_overlap.hip -o sched_overlap && ./sched_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// a workgroup that lives `iters` dependent L2/HBM round trips (+ a little math); LDS bytes and registers as template knobs
template <int THREADS, int LDS_FLOATS, int REGS>
__global__ __launch_bounds__(THREADS) void work_kernel(const v4f *__restrict__ src, size_t n_v4, int iters, float *sink)
{
    __shared__ float lds[LDS_FLOATS];
    v4f acc[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
    const size_t msk = n_v4 - 1;      // n_v4 is a power of two
    size_t i = ((size_t)blockIdx.x * THREADS + threadIdx.x) & msk;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const v4f a = src[(i + (size_t)r * 4099) & msk];
            acc[r] += a * a + a;
        }
        lds[threadIdx.x % LDS_FLOATS] = acc[0].x;
        __syncthreads();
        // the next address depends on loaded data (always 0 in practice: the buffer is zero-filled)
        i = (i + 7919 * THREADS + (size_t)(lds[(threadIdx.x + 1) % LDS_FLOATS] != 0.f)) & msk;
        __syncthreads();
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < REGS; ++r) t += acc[r].x + acc[r].y + acc[r].z + acc[r].w;
    if (t == 1.2345e-30f) sink[0] = t;
}

struct Ctx {
    v4f *buf; size_t n_v4; float *sink;
    int chain_len, chain_wgs, chain_iters, flood_len, flood_wgs, flood_iters;
    int big;      // chain workgroups: 0 = 256 threads / 24 KB LDS (forward-like), 1 = 512 threads / 64 KB LDS (backward-like)
};

static void launch_chain(const Ctx &c, hipStream_t s)
{
    for (int k = 0; k < c.chain_len; ++k) {
        if (c.big) hipLaunchKernelGGL((work_kernel<512, 16384, 6>), dim3(c.chain_wgs), dim3(512), 0, s, c.buf, c.n_v4, c.chain_iters, c.sink);
        else hipLaunchKernelGGL((work_kernel<256, 6144, 6>), dim3(c.chain_wgs), dim3(256), 0, s, c.buf, c.n_v4, c.chain_iters, c.sink);
    }
}
static void launch_flood(const Ctx &c, hipStream_t s)
{
    for (int k = 0; k < c.flood_len; ++k)
        hipLaunchKernelGGL((work_kernel<256, 8192, 32>), dim3(c.flood_wgs), dim3(256), 0, s, c.buf, c.n_v4, c.flood_iters, c.sink);
}

static int g_prio_hi = -1, g_prio_lo = 0;
static hipGraphExec_t capture(const Ctx &c, hipStream_t s, bool chain, bool flood, unsigned flags, bool node_prio)
{
    hipGraph_t g;
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    if (chain && flood) {
        CK(hipEventRecord(fork, s));
        CK(hipStreamWaitEvent(s2, fork, 0));
        launch_flood(c, s2);
        launch_chain(c, s);
        CK(hipEventRecord(join, s2));
        CK(hipStreamWaitEvent(s, join, 0));
    } else if (chain) launch_chain(c, s);
    else launch_flood(c, s);
    CK(hipStreamEndCapture(s, &g));
    if (node_prio) {
        size_t n = 0;
        CK(hipGraphGetNodes(g, nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n);
        CK(hipGraphGetNodes(g, nodes.data(), &n));
        int nset = 0;
        for (size_t i = 0; i < n; ++i) {
            hipGraphNodeType ty;
            CK(hipGraphNodeGetType(nodes[i], &ty));
            if (ty != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams p;
            CK(hipGraphKernelNodeGetParams(nodes[i], &p));
            const bool is_chain = (int)p.gridDim.x == c.chain_wgs;
            hipKernelNodeAttrValue v;
            v.priority = is_chain ? g_prio_hi : g_prio_lo;
            hipError_t e = hipGraphKernelNodeSetAttribute(nodes[i], hipKernelNodeAttributePriority, &v);
            if (e == hipSuccess) ++nset; else { printf("    (node priority attribute refused: %s)\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
        }
        printf("    (priority set on %d kernel nodes)\n", nset);
    }
    hipGraphExec_t ex;
    hipError_t e = hipGraphInstantiateWithFlags(&ex, g, flags);
    if (e != hipSuccess) { printf("    (instantiate with flags %u failed: %s; falling back to 0)\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); CK(hipGraphInstantiateWithFlags(&ex, g, 0)); }
    return ex;
}

struct Res { double chain_us, total_us; };

// runs `reps` times: fA() starts the flood side, fB() the chain side; events bracket the chain on sB and everything on both
template <class FA, class FB>
static Res measure(hipStream_t sA, hipStream_t sB, FA fA, FB fB, int reps, bool haveA, bool haveB)
{
    hipEvent_t b0, b1, a1, start;
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&start));
    double chain = 0, total = 0;
    for (int r = -2; r < reps; ++r) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(start, sA));
        CK(hipStreamWaitEvent(sB, start, 0));
        if (haveA) fA();
        CK(hipEventRecord(b0, sB));
        if (haveB) fB();
        CK(hipEventRecord(b1, sB));
        CK(hipStreamWaitEvent(sA, b1, 0));
        CK(hipEventRecord(a1, sA));
        CK(hipEventSynchronize(a1));
        float c_ms = 0, t_ms = 0;
        CK(hipEventElapsedTime(&c_ms, b0, b1));
        CK(hipEventElapsedTime(&t_ms, start, a1));
        if (r >= 0) { chain += c_ms * 1e3; total += t_ms * 1e3; }
    }
    return Res{chain / reps, total / reps};
}

int main(int argc, char **argv)
{
    Ctx c;
    c.n_v4 = (size_t)64 * 1024 * 1024 / 16;
    const int flood_iters = argc > 1 ? atoi(argv[1]) : 10, chain_it = argc > 2 ? atoi(argv[2]) : 6, flood_len = argc > 3 ? atoi(argv[3]) : 3;
    CK(hipMalloc(&c.buf, c.n_v4 * 16)); CK(hipMemset(c.buf, 0, c.n_v4 * 16)); CK(hipMalloc(&c.sink, 4));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    g_prio_hi = hi; g_prio_lo = lo;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int reps = 6;
    for (int big = 0; big < 2; ++big) {
        c.big = big;
        c.chain_len = 12; c.chain_wgs = big ? 1600 : 560; c.chain_iters = chain_it;
        c.flood_len = flood_len; c.flood_wgs = 4480; c.flood_iters = flood_iters;
        printf("\n=== chain of %d launches x %d workgroups of %s  beside  %d flood launches x %d workgroups of 256 threads\n", c.chain_len,
               c.chain_wgs, big ? "512 threads / 64 KB LDS" : "256 threads / 24 KB LDS", c.flood_len, c.flood_wgs);
        hipStream_t sA, sB, sBhi, sAm, sBm;
        CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
        CK(hipStreamCreateWithPriority(&sBhi, hipStreamNonBlocking, hi));
        // CU masks: the chain gets a quarter of the CUs, the flood the rest (bit i = CU i; 256 CUs = 8 words)
        uint32_t mB[8], mA[8];
        for (int w = 0; w < 8; ++w) { mB[w] = 0x03030303u; mA[w] = ~mB[w]; }
        hipError_t em1 = hipExtStreamCreateWithCUMask(&sBm, 8, mB), em2 = hipExtStreamCreateWithCUMask(&sAm, 8, mA);
        const bool masks = em1 == hipSuccess && em2 == hipSuccess;
        if (!masks) { printf("(CU-mask streams unavailable: %s)\n", hipGetErrorString(em1 != hipSuccess ? em1 : em2)); (void)hipGetLastError(); }

        auto eagerA = [&](hipStream_t s) { return [&c, s]() { launch_flood(c, s); }; };
        auto eagerB = [&](hipStream_t s) { return [&c, s]() { launch_chain(c, s); }; };
        Res r;
        r = measure(sA, sB, eagerA(sA), eagerB(sB), reps, false, true);  printf("chain alone (eager)                     chain %8.1f us\n", r.chain_us);
        r = measure(sA, sB, eagerA(sA), eagerB(sB), reps, true, false);  printf("flood alone (eager)                                         total %8.1f us\n", r.total_us);
        r = measure(sA, sB, eagerA(sA), eagerB(sB), reps, true, true);   printf("eager, two streams, normal priority     chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        r = measure(sA, sBhi, eagerA(sA), eagerB(sBhi), reps, true, true); printf("eager, chain stream HIGH priority       chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        if (masks) {
            r = measure(sAm, sBm, eagerA(sAm), eagerB(sBm), reps, false, true); printf("chain alone on its 64-CU mask (eager)   chain %8.1f us\n", r.chain_us);
            r = measure(sAm, sBm, eagerA(sAm), eagerB(sBm), reps, true, false); printf("flood alone on its 192-CU mask (eager)                      total %8.1f us\n", r.total_us);
            r = measure(sAm, sBm, eagerA(sAm), eagerB(sBm), reps, true, true);  printf("eager, CU-masked streams (64 / 192)     chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        }
        // graphs: one per side, launched on the two streams
        hipStream_t cap; CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        hipGraphExec_t gA = capture(c, cap, false, true, 0, false), gB = capture(c, cap, true, false, 0, false);
        auto graphOn = [&](hipGraphExec_t g, hipStream_t s) { return [g, s]() { CK(hipGraphLaunch(g, s)); }; };
        r = measure(sA, sB, graphOn(gA, sA), graphOn(gB, sB), reps, false, true);   printf("chain graph alone                       chain %8.1f us\n", r.chain_us);
        r = measure(sA, sB, graphOn(gA, sA), graphOn(gB, sB), reps, true, false);   printf("flood graph alone                                           total %8.1f us\n", r.total_us);
        r = measure(sA, sB, graphOn(gA, sA), graphOn(gB, sB), reps, true, true);    printf("two graphs, two streams, normal prio    chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        r = measure(sA, sBhi, graphOn(gA, sA), graphOn(gB, sBhi), reps, true, true); printf("two graphs, chain stream HIGH prio      chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        if (masks) {
            r = measure(sAm, sBm, graphOn(gA, sAm), graphOn(gB, sBm), reps, true, true); printf("two graphs on the CU-masked streams     chain %8.1f us  total %8.1f us\n", r.chain_us, r.total_us);
        }
        // one graph, two branches
        hipGraphExec_t gAB = capture(c, cap, true, true, 0, false);
        r = measure(sA, sB, graphOn(gAB, sA), graphOn(gAB, sA), reps, true, false);  printf("ONE graph, two branches                                     total %8.1f us\n", r.total_us);
        r = measure(sBhi, sB, graphOn(gAB, sBhi), graphOn(gAB, sBhi), reps, true, false); printf("ONE graph, two branches, on HIGH stream                     total %8.1f us\n", r.total_us);
        hipGraphExec_t gABp = capture(c, cap, true, true, hipGraphInstantiateFlagUseNodePriority, true);
        r = measure(sA, sB, graphOn(gABp, sA), graphOn(gABp, sA), reps, true, false); printf("ONE graph, node priorities (chain high)                     total %8.1f us\n", r.total_us);
        fflush(stdout);
    }
    return 0;
}
