// Micro-benchmark (tools only): per-CU rate of 16-byte global loads that hit L2.  Every workgroup streams REPS times over its own
// window of WIN bytes (mode 0: distinct windows, L2-resident; mode 1: all workgroups share one window, like a weight image).
//   hipcc --offload-arch=gfx950 -O3 l2_load_bw.hip -o l2_load_bw && ./l2_load_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const v4f *__restrict__ src, int win_v4, int reps, int shared, float *sink)
{
    const v4f *p = src + (shared ? 0 : (size_t)blockIdx.x * win_v4);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < win_v4; i += 4 * 256) {
            const v4f a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
            acc += (a + b) + (c + d);
        }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
}

int main()
{
    const int wins_kb[] = {16, 64, 128, 256};
    const int wgs_per_cu[] = {1, 2, 4, 8};
    float *sink;
    hipMalloc(&sink, 4);
    for (int shared = 0; shared < 2; ++shared)
        for (int wk : wins_kb)
            for (int wpc : wgs_per_cu) {
                const int nwg = 256 * wpc, win_v4 = wk * 1024 / 16, reps = 4096 * 16 / wk;
                v4f *buf;
                const size_t bytes = (size_t)(shared ? 1 : nwg) * wk * 1024;
                hipMalloc(&buf, bytes);
                hipMemset(buf, 0, bytes);
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(stream_kernel, dim3(nwg), dim3(256), 0, 0, buf, win_v4, 2, shared, sink);
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_kernel, dim3(nwg), dim3(256), 0, 0, buf, win_v4, reps, shared, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double tot = (double)nwg * wk * 1024.0 * reps;
                printf("%s window %4d KB  %d wg/CU  footprint %7.1f MB : %7.2f TB/s = %6.1f B/clk/CU @2.4GHz (%.3f ms)\n",
                       shared ? "shared  " : "distinct", wk, wpc, bytes / 1e6, tot / ms / 1e9, tot / (ms * 1e-3) / 2.4e9 / 256.0, ms);
                hipFree(buf);
            }
    return 0;
}
