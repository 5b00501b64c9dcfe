// Shared device helpers for libfcn_hip.so (gfx950 only: wave64, 32x32x2 f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fcn_hip.h"

#define FCN_WAVE 64

// dynamic LDS of a kernel (size given at launch); a macro so that the host emulation of tests/host_harness can map it
#ifndef FCN_DYN_LDS
#define FCN_DYN_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FCN_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define FCN_TRY(expr)                              \
    do {                                           \
        int rc__ = (int)(expr);                    \
        if (rc__ != 0) return rc__;                \
    } while (0)

// ---- single-instruction forms of staging arithmetic hipcc does not select on its own.  The operand path of the GEMMs (BatchNorm
// + ReLU + split encode per element, row addresses per load) is VALU-issue-bound: 157 vector instructions per six MFMAs in the FCN
// forward's K loop before these (tools/loophist.py on the ISA).  The host emulation of tests/ (no inline asm there: FCN_HOST_EMU)
// compiles the plain C++ restatement beside each.
// relu(y) where `keep` (+inf: the element counts) or 0 (it is padding): v_med3_f32 instead of v_max_f32 + v_cndmask.  A NaN comes
// out as 0, as fmaxf(NaN, 0) does (the producer has raised the workspace's sticky flag for it).
__device__ __forceinline__ float fcn_relu_keep(float y, float keep) { return __builtin_amdgcn_fmed3f(y, 0.f, keep); }
__device__ __forceinline__ float fcn_keep(bool ok) { return ok ? __builtin_inff() : 0.f; }
// a * b + c on the 24-bit integer multiplier (v_mad_u32_u24: full rate; the 32-bit multiplies hipcc emits for row addresses
// -- v_mad_u64_u32 -- are quarter rate).  a, b < 2^24 and the true result < 2^32; b wave-uniform (one SGPR operand).
__device__ __forceinline__ unsigned fcn_mad24(unsigned a, unsigned b, unsigned c)
{
#ifdef FCN_HOST_EMU
    return a * b + c;
#else
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
#endif
}

// a wave-uniform value as an opaque SGPR value (no code, not a scheduling barrier): hipcc cannot fold it back into a per-lane select
__device__ __forceinline__ int fcn_opaque_sgpr(int v)
{
#ifndef FCN_HOST_EMU
    asm("" : "+s"(v));
#endif
    return v;
}

// a 32-bit per-lane value as an opaque VGPR value (no code) at the point of use: a loop-invariant byte offset is otherwise
// zero-extended ONCE outside the loop and every load then adds a 64-bit VGPR pair to its base (v_lshl_add_u64) instead of taking
// the SGPR-base + 32-bit-VGPR-offset addressing mode
__device__ __forceinline__ unsigned fcn_opaque_v32(unsigned v)
{
#ifndef FCN_HOST_EMU
    asm("" : "+v"(v));
#endif
    return v;
}
// an LDS pointer as an opaque VGPR value (no code): two pointers that differ by a constant become two BASES for hipcc's
// read-merging pass, which pairs only reads off one base (gemm_tile.h mma_chunk).  The pointer keeps its LDS address space (a
// generic pointer behind an asm would be read with flat loads).
#ifdef FCN_HOST_EMU
typedef const uint32_t *fcn_lds_u32p;
__device__ __forceinline__ fcn_lds_u32p fcn_opaque_lds(const uint32_t *p) { return p; }
#else
typedef const uint32_t __attribute__((address_space(3))) *fcn_lds_u32p;
__device__ __forceinline__ fcn_lds_u32p fcn_opaque_lds(const uint32_t *p)
{
    fcn_lds_u32p q = (fcn_lds_u32p)p;
    asm("" : "+v"(q));
    return q;
}
#endif

// dy of a BatchNorm backward from (dz, y), the folded coefficients of fcn_bnbwd_coef and the entry's multiplicity w
__device__ __forceinline__ float fcn_bn_dy(float c0, float c1, float c2, float c3, float dz, float y, float w)
{
    return fmaf(c0, dz, -(w * fmaf(c2, y - c1, c3)));
}
__device__ __forceinline__ float fcn_bn_dy1(float c0, float c1, float c2, float c3, float dz, float y)      // w = 1
{
    return fmaf(c0, dz, -fmaf(c2, y - c1, c3));
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// REPLICATED statistic slots.  The BatchNorm sums (forward: sum / sum of squares; backward: sum dz / sum dz*xhat) are fp64
// atomics of per-workgroup partials.  Atomics on ONE address are served one after the other by the memory side (~18 ns each on
// MI355X): a launch of 600 workgroups that all add to the same 128 channel slots spent 5-11 us of a 14-17 us kernel there
// (timing builds without the atomics: PointNet conv3 of the narrow scales 16.9 -> 11.5 us, their data-gradient kernels 23 us of
// 93 per scale).  Every slot therefore exists FCN_STAT_REP times; a workgroup adds to replica blockIdx.x % FCN_STAT_REP (its XCD)
// and every consumer sums the replicas in replica order (fcn_rep_sum).  A replica block is `stride` doubles apart.
#ifndef FCN_STAT_REP
#define FCN_STAT_REP 8
#endif
__device__ __forceinline__ double fcn_rep_sum(const double *p, int stride)
{
    double v = p[0];
#pragma unroll
    for (int r = 1; r < FCN_STAT_REP; ++r) v += p[(int64_t)r * stride];
    return v;
}
// Consumer-side BatchNorm finalisation reads 2 x REP sums per channel in every workgroup's prologue.  The forward GEMM requests them
// FIRST -- ahead of the dependent round trips of the tile lookup, with a compiler fence (FCN_LOAD_FENCE) that keeps them there --
// so their latency runs beside that chain instead of after it: tools/pn_probe.py fwd measured the prologue at 22-40 % of a conv3
// workgroup's cycles before and 9-15 % after (workgroup cycles -17...-27 %; forward of a scale alone -2.4 us; the whole step: within
// noise).  The same move in the data-gradient GEMMs (up to 76 loads per thread in front of the lookup) measured 1.5 % SLOWER over
// the step and is not in the tree.  FCN_EARLY_STATS=0 (tuning builds) keeps the loads at the point of use.
#ifndef FCN_EARLY_STATS
#define FCN_EARLY_STATS 1
#endif
#define FCN_LOAD_FENCE() asm volatile("" ::: "memory")
__device__ __forceinline__ int fcn_rep_id() { return (int)(blockIdx.x % FCN_STAT_REP); }

// Max-pool keys (pooling folded into conv3's epilogue, pointnet_fwd.hip).  BatchNorm + ReLU is monotone in the conv output y
// with the sign of gamma (rstd > 0), so the row that wins max_r relu(bn(y_r)) is the row with the largest sgn(gamma) * y_r, known
// before the batch statistics are: a 64-bit key orders (value, then EARLIER row) under an unsigned max.  0 = "no row".
//   high word: the fp32 bits of sgn(gamma) * y mapped to an order-preserving unsigned; low word: ~row.
__device__ __forceinline__ float fcn_pool_orient(float v, float gamma) { return gamma > 0.f ? v : (gamma < 0.f ? -v : 0.f); }
__device__ __forceinline__ unsigned long long fcn_pool_key(float oriented, int row)
{
    const unsigned b = __float_as_uint(oriented);
    const unsigned o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);
}
// -> y of the winning row (as conv3 stored it) and the row, from the key's two words (kept as 32-bit values: ROCm 7.2's
// instruction selection crashes on the 64-bit shift / compare form of this function)
__device__ __forceinline__ float fcn_pool_key_value(unsigned hi, unsigned lo, float gamma, int &row)
{
    const unsigned b = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
    row = (int)(0xffffffffu - lo);
    const float s = __uint_as_float(b);
    return gamma < 0.f ? -s : s;
}

// Coefficients of a BatchNorm backward, derived by every CONSUMER workgroup from the batch sums (a few fp64 products per
// channel) instead of by a one-workgroup launch between two layers of a latency-bound chain.  With xhat = (y - mean) * rstd,
//   dy = gamma*rstd * (dz - w * (sum(dz)/M + xhat * sum(dz*xhat)/M))             (w = 1 outside the PointNet's entry space)
// is evaluated per element in its FOLDED form -- a subtraction, two fused multiply-adds and (entry space) one product:
//   dy = fma(c0, dz, -w * fma(c2, y - c1, c3))
//   c0 = gamma * rstd, c1 = mean, c2 = c0 * rstd * sum(dz*xhat) / M, c3 = c0 * sum(dz) / M        (c4 unused, 0)
// (the literal form costs six operations per element in the staging path of every backward GEMM, which is vector-issue-bound;
// the folded one has the same terms of the same magnitudes -- no new cancellation -- and the per-channel products are formed in fp64)
// bstat: sum dz [C], sum dz * xhat [C] (final when the consumer starts); bn: scale, shift, mean, rstd [C] of the forward.
struct FcnBnBwd {
    const double *bstat;       // replica 0 of [sum dz [C], sum dz * xhat [C]]
    int rep_stride;            // doubles between replicas (fcn_rep_sum)
    const float *gamma, *bn;
    double invM;
    float *dgamma, *dbeta;     // exported by the workgroup flagged `pub` (fp32), or null
};
__device__ __forceinline__ void fcn_bnbwd_coef(const FcnBnBwd &q, int C, int c, float (&cf)[5], bool pub)
{
    const double db = fcn_rep_sum(q.bstat + c, q.rep_stride), dg = fcn_rep_sum(q.bstat + C + c, q.rep_stride);
    const float rstd = q.bn[3 * C + c];
    cf[0] = q.gamma[c] * rstd;
    cf[1] = q.bn[2 * C + c];
    const double c0 = (double)cf[0];
    cf[2] = (float)(c0 * (double)rstd * (dg * q.invM));
    cf[3] = (float)(c0 * (db * q.invM));
    cf[4] = 0.f;
    if (pub && q.dgamma) { q.dgamma[c] = (float)dg; q.dbeta[c] = (float)db; }
}

// 1 / sqrt(x) in fp64 from the fp32 rsqrt + three Newton steps (full double accuracy for x > 0 within float range -- a
// variance + eps): a dozen fp64 operations instead of the software sqrt + division sequences (~100 instructions)
__device__ __forceinline__ double fcn_rsqrt64(double x)
{
    double y = (double)rsqrtf((float)x);
#pragma unroll
    for (int i = 0; i < 3; ++i) y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Layout of fcn_pn_ws.stat (doubles)
//   [0]      sum w          (= B*L*K)
//   [1..3]   sum w*u
//   [4..9]   sum w*u*u^T    (xx,xy,xz,yy,yz,zz)
//   then FCN_STAT_REP replica blocks of 2*C2 + 2*C3 doubles, block r at 16 + r * (2*C2 + 2*C3):
//     [0 .. 2*C2)            layer-2 sum, sumsq
//     [2*C2 .. 2*C2+2*C3)    layer-3 sum, sumsq
// fcn_pn_ws.bstat: FCN_STAT_REP replica blocks of 2*C3 + 2*C2 + 4*C1 doubles (dbeta3, dgamma3, dbeta2, dgamma2, Q[4][C1])
#define FCN_STAT_MOM 0
#define FCN_STAT_L2 16

// Layout of fcn_pn_ws.bn (floats): layer j block at offset 4*(sum of previous widths):
//   scale[C], shift[C], mean[C], rstd[C]
__host__ __device__ inline int fcn_bn_off(int layer, int C1, int C2) {
    return layer == 0 ? 0 : (layer == 1 ? 4 * C1 : 4 * (C1 + C2));
}
