// Micro-benchmark (tools only): the per-launch floor of a chain of DEPENDENT small launches -- eager, and as a replayed hipGraph
// captured on one stream or ping-ponging between the origin stream and 1-3 forked ones.  Result on MI355X / ROCm 7.2: 2.82 us
// eager, 2.06 us per launch in a graph, and the SAME 2.06 us whichever streams the chain was captured on -- the graph executor
// places nodes by the graph's shape (a linear chain stays on one internal stream), not by the streams of the capture, so this
// cannot measure a cross-stream release; the ConvFeatNet chains (12-13 launches) pay this floor per launch.
//   hipcc --offload-arch=gfx950 -O3 chain_hop.hip -o chain_hop && ./chain_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void work(float *p, int iters)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0000001f, 1e-9f);      // a dependent chain: ~4 cycles per iteration
    p[i] = v;
}

static double run_graph(int nstream, int nlaunch, int grid, int iters, float *buf)
{
    hipStream_t s[4], origin;
    hipEvent_t ev[64];
    CK(hipStreamCreate(&origin));
    for (int i = 0; i < 4; ++i) CK(hipStreamCreate(&s[i]));
    for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
    int prev = -1;                                  // -1: origin
    for (int l = 0; l < nlaunch; ++l) {
        // (ROCm 7.2 stream capture crashes on a fork from an already-forked stream: the chain returns to the origin stream
        // between two forked ones -- 2: origin, s0, origin, s0 ...; 3: origin, s0, origin, s1, ...)
        const int cur = nstream == 1 ? -1 : ((l & 1) ? ((l >> 1) % (nstream - 1)) : -1);
        hipStream_t sp = prev < 0 ? origin : s[prev], sc = cur < 0 ? origin : s[cur];
        if (sc != sp) { CK(hipEventRecord(ev[l], sp)); CK(hipStreamWaitEvent(sc, ev[l], 0)); }
        hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, sc, buf, iters);
        prev = cur;
    }
    if (prev >= 0) { CK(hipEventRecord(ev[63], s[prev])); CK(hipStreamWaitEvent(origin, ev[63], 0)); }
    CK(hipStreamEndCapture(origin, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, origin));
    CK(hipStreamSynchronize(origin));
    const int R = 200;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < R; ++i) CK(hipGraphLaunch(ge, origin));
    CK(hipStreamSynchronize(origin));
    auto t1 = std::chrono::high_resolution_clock::now();
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    for (int i = 0; i < 4; ++i) CK(hipStreamDestroy(s[i]));
    CK(hipStreamDestroy(origin));
    for (int i = 0; i < 64; ++i) CK(hipEventDestroy(ev[i]));
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / R / nlaunch;
}

static double run_eager(int nlaunch, int grid, int iters, float *buf)
{
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, st, buf, iters);
    CK(hipStreamSynchronize(st));
    const int R = 200;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < R * nlaunch; ++i) hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, st, buf, iters);
    CK(hipStreamSynchronize(st));
    auto t1 = std::chrono::high_resolution_clock::now();
    CK(hipStreamDestroy(st));
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / R / nlaunch;
}

int main()
{
    float *buf; CK(hipMalloc(&buf, (size_t)4480 * 256 * 4)); CK(hipMemset(buf, 0, (size_t)4480 * 256 * 4));
    const int NL = 24;
    printf("us per launch of a %d-launch dependent chain (replayed graph unless eager):\n", NL);
    printf("%-34s %8s %8s %8s %8s %8s\n", "kernel", "eager", "1 stream", "origin+1", "origin+2", "origin+3");
    const int grids[3] = {560, 560, 1960}, iters[3] = {1, 2000, 4000};
    for (int k = 0; k < 3; ++k) {
        char nm[64]; snprintf(nm, sizeof nm, "%d wg x 256, %d dependent fma", grids[k], iters[k]);
        printf("%-34s %8.2f", nm, run_eager(NL, grids[k], iters[k], buf));
        for (int ns = 1; ns <= 4; ++ns) printf(" %8.2f", run_graph(ns, NL, grids[k], iters[k], buf));
        printf("\n");
    }
    return 0;
}
