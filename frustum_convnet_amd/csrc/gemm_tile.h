// 32x32x2 f32 MFMA tile machinery shared by the forward / dgrad / wgrad kernels.
//
// Workgroup = 256 threads = 4 waves arranged 2 (M) x 2 (N); each wave owns MT x NT MFMA tiles of
// 32x32, i.e. the workgroup tile is (64*MT) x (64*NT).  Operands are staged k-major in LDS
// ([k][m] and [k][n], leading dimension padded to an odd number of dwords) so that the MFMA operand
// reads -- lane l needs A[m0 + (l&31)][k + (l>>5)] and B[k + (l>>5)][n0 + (l&31)] -- are 32 consecutive
// dwords per half-wave: conflict-free ds_read_b32.  v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise an
// fmaf chain, cdna guide section 3), which is what keeps the logits within 1e-4 of the fp32 reference.
#pragma once
#include "fcn_common.h"

// Native 4-wide vectors for register staging.  HIP's float4 is a struct: copying an array element of it between
// address spaces (global -> register array -> LDS) is lowered to memcpy through a PRIVATE (scratch) array that SROA
// does not always remove -- v4f / v4i are plain LLVM vectors and stay in VGPRs.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
// loads through an explicit GLOBAL address space: a pointer whose provenance the compiler lost (select between
// kernel-struct fields, inline-asm pin) would otherwise be generic and load with flat_load, which counts on lgkmcnt
// as well -- the next s_waitcnt lgkmcnt(0) in front of an MFMA would then also wait for the prefetch.
typedef const v4f __attribute__((address_space(1))) *gv4fp;
typedef const v4i __attribute__((address_space(1))) *gv4ip;
__device__ __forceinline__ v4f ldg4(const float *p) { return *(gv4fp)p; }
__device__ __forceinline__ v4i ldg4i(const int *p) { return *(gv4ip)p; }
__device__ __forceinline__ void sts4(float *p, v4f v) { *(v4f *)p = v; }
__device__ __forceinline__ v4f zero4() { v4f z = {0.f, 0.f, 0.f, 0.f}; return z; }

#define GT 256          // threads per workgroup
#define KC 32           // reduction chunk staged per iteration

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// One KC-deep chunk: As is [KC][LDA] (m fastest), Bs is [KC][LDB] (n fastest).
template <int MT, int NT, int LDA, int LDB>
__device__ __forceinline__ void mma_chunk(const float *As, const float *Bs, int arow0, int bcol0,
                                          f32x16 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, lh = lane >> 5;
    const float *ap = As + lh * LDA + arow0 + l31;
    const float *bp = Bs + lh * LDB + bcol0 + l31;
    // operands of k-step kk+2 are fetched from LDS before the MFMAs of k-step kk issue, so the ds_read latency
    // hides behind 4 x 64 cycles of matrix work instead of stalling every step on lgkmcnt(0)
    float a[2][MT], b[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = ap[i * 32];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[0][j] = bp[j * 32];
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < KC) {
#pragma unroll
            for (int i = 0; i < MT; ++i) a[nxt][i] = ap[(kk + 2) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[nxt][j] = bp[(kk + 2) * LDB + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE this step's MFMAs (hipcc otherwise sinks it)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Row (M index inside the 32x32 tile) held by accumulator register `reg` of this lane.
__device__ __forceinline__ int acc_row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }

// conv1 + BN1 + ReLU of one entry, folded: alpha = scale * W1 row, t = shift.  The SAME expression is
// used by the forward, the dgrad ReLU mask and the wgrad operand so the three agree bit-for-bit.
__device__ __forceinline__ float l1_pre(const float *al3, float t, float ux, float uy, float uz) {
    return fmaf(al3[0], ux, fmaf(al3[1], uy, fmaf(al3[2], uz, t)));
}
