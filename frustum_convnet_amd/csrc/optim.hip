// Adam over ONE flat fp32 buffer: the optimiser step of the reference's loop (train/train_net_det.py:321-339 builds
// optim.Adam(lr, weight_decay); :131-133 calls it every iteration).  The 79 parameter tensors, their gradients and
// both moments live in four contiguous 13.3 MB buffers, so the step is a single streaming kernel (4 reads + 3 writes
// per element, HBM-bound) instead of a multi-tensor launch chain, and the same flat gradient is what one RCCL
// all-reduce exchanges.  Arithmetic follows torch.optim.Adam (L2 weight decay folded into the gradient, lerp form of
// the first moment, eps added after the bias-corrected sqrt).
#include "fcn_common.h"
#include "gemm_tile.h"

struct AdamArgs {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
    const float *hyper;        // device: lr, beta1, beta2, eps, weight_decay, grad_scale
    int64_t *step;             // device step counters, ONE PER WORKGROUP (each reads and advances its own slot: no
                               // cross-workgroup hazard and no same-address atomic -- a ticket counter taken by
                               // 1600 workgroups serialises in L2 and cost 60 us of a 70 us launch)
};

#define ADAM_T 256
#define ADAM_V 2               // 4-vectors per thread

// moments of one element / one 4-vector (T = float or v4f); the parameter update needs a per-lane sqrt and follows
template <class T>
__device__ __forceinline__ void adam_moments(T p, T g, T &m, T &v, float b1, float b2, float wd, float gs)
{
    g = wd * p + g * gs;
    m = m + (g - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * g * g;
}

__global__ __launch_bounds__(ADAM_T) void adam_kernel(AdamArgs a)
{
    __shared__ float bcs[2];
    const float lr = a.hyper[0], b1 = a.hyper[1], b2 = a.hyper[2], eps = a.hyper[3], wd = a.hyper[4], gs = a.hyper[5];
    if (threadIdx.x == 0) {             // the two fp64 pow() are ~600 instructions: once per workgroup, not per thread
        const double t = (double)(a.step[blockIdx.x] + 1);
        bcs[0] = (float)(1.0 - pow((double)b1, t));
        bcs[1] = (float)(1.0 - pow((double)b2, t));
    }
    const int64_t n4 = a.n >> 2;
    // ADAM_V 4-vectors per thread, every load issued before the first use
    const int64_t base = (int64_t)blockIdx.x * (ADAM_T * ADAM_V) + threadIdx.x;
    v4f p[ADAM_V], g[ADAM_V], m[ADAM_V], v[ADAM_V];
#pragma unroll
    for (int q = 0; q < ADAM_V; ++q) {
        const int64_t i = min(base + q * ADAM_T, n4 - 1);
        p[q] = ldg4(a.p + 4 * i); g[q] = ldg4(a.g + 4 * i); m[q] = ldg4(a.m + 4 * i); v[q] = ldg4(a.v + 4 * i);
    }
    __syncthreads();
    const float lr_c = lr / bcs[0], rsq_bc2 = 1.f / sqrtf(bcs[1]);
#pragma unroll
    for (int q = 0; q < ADAM_V; ++q) {
        const int64_t i = base + q * ADAM_T;
        if (i < n4) {
            adam_moments(p[q], g[q], m[q], v[q], b1, b2, wd, gs);
            const v4f rt = {sqrtf(v[q].x), sqrtf(v[q].y), sqrtf(v[q].z), sqrtf(v[q].w)};
            p[q] = p[q] - lr_c * (m[q] / (rt * rsq_bc2 + eps));
            sts4(a.p + 4 * i, p[q]); sts4(a.m + 4 * i, m[q]); sts4(a.v + 4 * i, v[q]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {           // tail (n not a multiple of 4)
        const int64_t i = (n4 << 2) + threadIdx.x;
        float pp = a.p[i], mm = a.m[i], vv = a.v[i];
        adam_moments(pp, a.g[i], mm, vv, b1, b2, wd, gs);
        pp = pp - lr_c * (mm / (sqrtf(vv) * rsq_bc2 + eps));
        a.p[i] = pp; a.m[i] = mm; a.v[i] = vv;
    }
    if (threadIdx.x == 0) a.step[blockIdx.x] += 1;     // thread 0 read it above
}

static int64_t adam_blocks(int64_t n) { const int64_t per = (int64_t)ADAM_T * ADAM_V; return ((n >> 2) + per - 1) / per; }

extern "C" int64_t fcn_adam_step_slots(int64_t n) { return n < 4 ? 0 : adam_blocks(n); }

extern "C" int fcn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 const float *hyper6, int64_t *step_slots, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper6 || !step_slots || n <= 0) return FCN_E_BADARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return FCN_E_BADARG;
    AdamArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.hyper = hyper6; a.step = step_slots;
    if (n < 4) return FCN_E_BADARG;
    const int64_t blocks = adam_blocks(n);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(ADAM_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The 'sgd' branch of the step loop (train/train_net_det.py:325-327: optim.SGD(lr, momentum, weight_decay)), same flat buffers.
// torch.optim.SGD arithmetic with dampening 0: g' = wd * p + g * grad_scale; buf = momentum * buf + g'; p -= lr * buf.  (torch
// seeds the buffer with the first gradient: identical to the recurrence from a zero buffer, so there is no step counter.)
__global__ __launch_bounds__(ADAM_T) void sgd_kernel(float *p, const float *g, float *buf, int64_t n, const float *hyper)
{
    const float lr = hyper[0], mu = hyper[1], wd = hyper[2], gs = hyper[3];
    const int64_t n4 = n >> 2, i = (int64_t)blockIdx.x * ADAM_T + threadIdx.x;
    if (i < n4) {
        const v4f pp = ldg4(p + 4 * i);
        const v4f b = mu * ldg4(buf + 4 * i) + (wd * pp + ldg4(g + 4 * i) * gs);
        sts4(buf + 4 * i, b);
        sts4(p + 4 * i, pp - lr * b);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        const float b = mu * buf[j] + (wd * p[j] + g[j] * gs);
        buf[j] = b;
        p[j] -= lr * b;
    }
}

extern "C" int fcn_sgd_step_f32(float *param, const float *grad, float *momentum_buf, int64_t n, const float *hyper4,
                                void *stream)
{
    if (!param || !grad || !momentum_buf || !hyper4 || n < 4) return FCN_E_BADARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) return FCN_E_BADARG;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)(((n >> 2) + ADAM_T - 1) / ADAM_T)), dim3(ADAM_T), 0, (hipStream_t)stream,
                       param, grad, momentum_buf, n, hyper4);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Measurement aid: one thread stores the GPU's constant-rate wall clock (100 MHz) into *slot.  Launched between the
// phases of a step it gives phase boundaries INSIDE a replayed hipGraph without a profiler attached
// (tools/phase_stamps.py); rocprofv3's kernel trace perturbs the overlap of the captured branches.
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }

extern "C" int fcn_stamp(uint64_t *slot, void *stream)
{
    if (!slot) return FCN_E_BADARG;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)slot);
    FCN_CHECK_LAUNCH();
    return 0;
}

// Which hipGraph capture `stream` is part of: *id = 0 when it is not capturing, else the runtime's capture id + 1.  Asked through
// THIS library so that the answer comes from the HIP runtime the kernels are launched on (a second libamdhip64 dlopen'ed by name
// would not know the stream): the Python layer ties state created during a capture -- a prefetched front -- to that capture.
extern "C" int fcn_stream_capture_id(void *stream, uint64_t *id)
{
    if (!id) return FCN_E_BADARG;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &status, &cid);
    if (e != hipSuccess) return (int)e;
    *id = status == hipStreamCaptureStatusActive ? (uint64_t)cid + 1u : 0u;
    return status == hipStreamCaptureStatusInvalidated ? FCN_E_BADARG : 0;
}
