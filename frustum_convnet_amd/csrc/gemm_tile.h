// MFMA tile machinery shared by the PointNet forward / dgrad / wgrad kernels and (operand encoding, fragment reads) the FCN.
//
// Workgroup = 256 threads = 4 waves arranged 2 (M) x 2 (N); each wave owns MT x NT MFMA tiles of
// 32x32, i.e. the workgroup tile is (64*MT) x (64*NT).  Operands are staged k-major in LDS
// ([k][m] and [k][n], leading dimension padded to an odd number of dwords) so that the MFMA operand
// reads are 32 consecutive dwords per half-wave: conflict-free ds_read_b32.
// Default arithmetic (MM_F16X3 / MM_BF16X3 below): every fp32 operand is split into two 16-bit parts while it is staged and a
// product is three v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation -- fp32-class results (logits within 1e-4 of the
// fp32 reference, parity-tested) at 5.3x the fp32 matrix rate.  MM_F32 (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain)
// is kept as the A/B reference mode, MM_BF16X1 as the throughput mode.
#pragma once
#include "fcn_common.h"
#include "fcn_tuning.h"

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

// Whether a scale's forward takes its max-pool from the keys of conv3's epilogue (pool_keys_kernel, pointnet_fwd.hip) -- asked by the
// forward AND by the backward, whose first kernel then finds the winners' pre-BN values in the routed-gradient buffer
static inline bool fcn_pn_key_pool(const fcn_pn_desc *d, const fcn_pn_ws *ws, int C3)
{
    return FCN_POOL_FUSED && d->nlc && ws->pkey && ws->ewin && C3 % 2 == 0 && C3 <= 2 * GT && (2 * GT) % C3 == 0;
}
#define KC 32           // reduction chunk staged per iteration

// ------------------------------------------------------------------------------------------------
// Matrix-core operand modes.  fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of the 16-bit MFMA
// rate on gfx950, and there is no xf32/TF32 form (cdna guide section 3).  The default mode therefore SPLITS every fp32
// operand x into two 16-bit parts hi = rne16(x), lo = rne16(x - hi) and forms a.b as a_hi.b_lo + a_lo.b_hi + a_hi.b_hi
// on v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation: 3 instructions per K=16 step instead of 8 x K=2 fp32 steps at
// twice the issue cost each -- 5.3x the fp32 matrix rate.
//   MM_F16X3  forward GEMMs: fp16 parts, 22 significand bits kept, error ~2^-21 per product (operands are O(1)
//             activations after BN+ReLU and weights; |x| must stay below 65504 -- an overflow shows up as inf/NaN);
//   MM_BF16X3 backward GEMMs: bf16 parts (fp32 exponent range -- gradients span many decades), error ~2^-17 per product;
//   MM_BF16X1 bf16 operands, the hi product only (FCN_PREC_BF16_OPS); MM_BF16S the same with bf16 storage of the big
//             intermediates -- BASELINE config 2's "bf16" throughput mode (FCN_PREC_BF16);
//   MM_F32    exact fp32 MFMA (bitwise an fmaf chain), kept as the reference mode for A/B runs.
// Measured on the CPU emulation of the whole net (tools/split_emulation.py): logits |err| vs fp64 9e-6 (f32) / 1.3e-5
// (f16x3) / 3e-4 (bf16x3: misses the 1e-4 bar, hence fp16 forwards) / 0.13 (bf16x1); parameter gradients with a bf16x3
// backward sit on the fp32 noise floor (2.0e-4 of max in both).
#define MM_F32 0
#define MM_F16X3 1
#define MM_BF16X3 2
#define MM_BF16X1 3
#define MM_BF16S 4          // MM_BF16X1 operands + the big intermediate tensors STORED as bf16 (St below)

template <int MM>
inline constexpr bool mm_x1 = (MM == MM_BF16X1 || MM == MM_BF16S);        // one bf16 MFMA per product
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int acc_row_c(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS operand layout, all modes: k-major, element (k, m) at [k * LD + m].  In the split modes two reduction-adjacent
// values (k even, k + 1) of one row/column share two dwords: row k holds the packed HI parts {hi(x_k), hi(x_k+1)},
// row k + 1 the packed LO parts -- exactly the register image of a 32x32x16 operand (lane (l&31, l>>5) holds
// k = 8*(l>>5) .. +7 as four dwords), so the MFMA loop reads its operands with plain ds_read_b32 and no VALU.
// enc2 turns such a pair into the two dwords to store.  fp16 parts: THREE instructions per pair -- v_cvt_pk_f16_f32, then
// v_fma_mixlo_f16 / v_fma_mixhi_f16, which read the fp16 hi part and the fp32 value, form x - hi in fp32 (exact) and round it to
// fp16 into one half of the destination: the same bits as the unpack / subtract / convert sequence hipcc emits for the C++ form
// (six instructions per pair).  bf16 parts: v_cvt_pk_bf16_f32, two masks, two subtractions, v_cvt_pk_bf16_f32.
template <int MM>
__device__ __forceinline__ void enc2(float x0, float x1, float &o0, float &o1)
{
    if constexpr (MM == MM_F32) {
        o0 = x0; o1 = x1;
    } else if constexpr (MM == MM_F16X3) {
        const f32x2 x = {x0, x1};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        o0 = __builtin_bit_cast(float, h);
#ifdef FCN_HOST_EMU
        const f32x2 r = {x0 - (float)h[0], x1 - (float)h[1]};          // exact
        const f16x2 l = __builtin_convertvector(r, f16x2);
        o1 = __builtin_bit_cast(float, l);
#else
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        uint32_t lb;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(x1));
        o1 = __builtin_bit_cast(float, lb);
#endif
    } else {
        const f32x2 x = {x0, x1};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        o0 = __builtin_bit_cast(float, hb);
        if constexpr (MM == MM_BF16X3) {
            const f32x2 r = {x0 - __builtin_bit_cast(float, hb << 16), x1 - __builtin_bit_cast(float, hb & 0xffff0000u)};
            const bf16x2 l = __builtin_convertvector(r, bf16x2);
            o1 = __builtin_bit_cast(float, l);
        } else {
            o1 = 0.f;
        }
    }
}
// four values along the reduction index (one float4 of a row): dwords for rows k .. k+3
template <int MM>
__device__ __forceinline__ void enc4(float x0, float x1, float x2, float x3, float (&o)[4])
{
    enc2<MM>(x0, x1, o[0], o[1]);
    enc2<MM>(x2, x3, o[2], o[3]);
}
// two float4 of reduction-adjacent rows (same four columns): the hi row and the lo row to store
template <int MM>
__device__ __forceinline__ void enc2x4(v4f a, v4f b, v4f &hi, v4f &lo)
{
    float h[4], l[4];
    enc2<MM>(a.x, b.x, h[0], l[0]); enc2<MM>(a.y, b.y, h[1], l[1]);
    enc2<MM>(a.z, b.z, h[2], l[2]); enc2<MM>(a.w, b.w, h[3], l[3]);
    hi.x = h[0]; hi.y = h[1]; hi.z = h[2]; hi.w = h[3];
    lo.x = l[0]; lo.y = l[1]; lo.z = l[2]; lo.w = l[3];
}

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

template <int MM>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c)
{
    if constexpr (MM == MM_F16X3)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One KCH-deep chunk (KC unless stated) of the weight-gradient GEMMs, whose reduction runs over ROWS of row-major operands: As is
// [KCH][LDA] (m fastest), Bs is [KCH][LDB] (n fastest), "pair-plane" order -- the two reduction-adjacent rows (2p, 2p + 1) of pair p
// sit at LDS row p (split modes: the packed HI parts of the pair; MM_F32: row 2p) and at LDS row KCH/2 + p (the packed LO parts;
// MM_F32: row 2p + 1).  mma_row_hi / mma_row_lo give the two LDS rows to the staging code.  The four dwords of one MFMA operand
// part are then NEIGHBOURING LDS rows: hipcc merges neighbouring reads into ds_read2_b32 and its two results are the two
// registers the operand wants next to each other (with the hi and lo row of a pair interleaved -- rows 2p, 2p + 1 -- every merged
// read delivered one register each to two different operands: 50 v_mov_b32 per 24 MFMAs in the PointNet weight-gradient loop).
template <int KCH>
__device__ __forceinline__ constexpr int mma_row_hi(int pair) { return pair; }
template <int KCH>
__device__ __forceinline__ constexpr int mma_row_lo(int pair) { return KCH / 2 + pair; }
template <int MM, int MT, int NT, int LDA, int LDB, int KCH = KC>
__device__ __forceinline__ void mma_chunk(const float *As, const float *Bs, int arow0, int bcol0,
                                          f32x16 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, lh = lane >> 5;
    if constexpr (MM == MM_F32) {
        // k-step kk (even) multiplies rows kk + lh: LDS row kk/2 of plane lh
        const float *ap = As + lh * (KCH / 2) * LDA + arow0 + l31;
        const float *bp = Bs + lh * (KCH / 2) * LDB + bcol0 + l31;
        // operands of k-step kk+2 are fetched from LDS before the MFMAs of k-step kk issue, so the ds_read latency
        // hides behind 4 x 64 cycles of matrix work instead of stalling every step on lgkmcnt(0)
        float a[2][MT], b[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[0][i] = ap[i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[0][j] = bp[j * 32];
#pragma unroll
        for (int kk = 0; kk < KCH; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < KCH) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = ap[((kk + 2) >> 1) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[nxt][j] = bp[((kk + 2) >> 1) * LDB + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE this step's MFMAs (hipcc otherwise sinks it)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // 32x32x16: lane (l31, lh) supplies k = 8*lh + 0..7 of row/column l31 -- the pairs 4*lh + {0,1,2,3} of the step: four
        // neighbouring LDS rows of the hi plane, four of the lo plane, each a conflict-free read of 32 consecutive dwords per half-wave
        // (one opaque base per 32-column block: off a COMMON base hipcc pairs the reads of two blocks' same row -- 32 dwords apart,
        // nearer than the next row -- and again every merged read feeds two operands)
        fcn_lds_u32p ap[MT], bp[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) ap[i] = fcn_opaque_lds((const uint32_t *)As + (4 * lh) * LDA + arow0 + l31 + i * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) bp[j] = fcn_opaque_lds((const uint32_t *)Bs + (4 * lh) * LDB + bcol0 + l31 + j * 32);
        constexpr bool X3 = !mm_x1<MM>;
#pragma unroll
        for (int ks = 0; ks < KCH; ks += 16) {
            u32x4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah[i][q] = ap[i][mma_row_hi<KCH>(ks / 2 + q) * LDA];
                    if constexpr (X3) al[i][q] = ap[i][mma_row_lo<KCH>(ks / 2 + q) * LDA];
                }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bh[j][q] = bp[j][mma_row_hi<KCH>(ks / 2 + q) * LDB];
                    if constexpr (X3) bl[j][q] = bp[j][mma_row_lo<KCH>(ks / 2 + q) * LDB];
                }
            // small cross terms first, then the hi.hi product; term-major so consecutive MFMAs hit different accumulators
            if constexpr (X3) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(ah[i], bl[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(al[i], bh[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(ah[i], bh[j], acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// STORAGE type of the big intermediate tensors (PointNet y2 / y3 / dy3 / dz2, the FCN's y / dz arenas).  fp32 in the split
// and f32 operand modes (and in FCN_PREC_BF16_OPS); in the bf16 throughput mode (FCN_PREC_BF16, BASELINE config 2: MM_BF16S)
// they are stored as bf16 -- half the
// HBM bytes of the step's dominant streams -- while accumulators, BatchNorm sums, pooled features, logits and every
// parameter gradient stay fp32.  The buffers keep their float-typed pointers and element counts (the caller sizes them for
// fp32; bf16 uses the first half): all index arithmetic is in ELEMENTS and these helpers scale it.
template <int MM>
struct St {
    static constexpr bool half = (MM == MM_BF16S);
    static constexpr int bytes = half ? 2 : 4;
};
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef const bf16x4 __attribute__((address_space(1))) *gbf4p;
template <int MM>
__device__ __forceinline__ v4f lds4e(const float *base, int64_t e)            // 4 consecutive stored elements -> fp32
{
    if constexpr (St<MM>::half) {
        const bf16x4 h = *(gbf4p)((const __bf16 *)base + e);
        return __builtin_convertvector(h, v4f);
    } else {
        return ldg4(base + e);
    }
}
template <int MM>
__device__ __forceinline__ void sts4e(float *base, int64_t e, v4f v)
{
    if constexpr (St<MM>::half) *(bf16x4 *)((__bf16 *)base + e) = __builtin_convertvector(v, bf16x4);
    else sts4(base + e, v);
}
template <int MM>
__device__ __forceinline__ float lds1e(const float *base, int64_t e)
{
    if constexpr (St<MM>::half) return (float)((const __bf16 *)base)[e];
    else return base[e];
}
template <int MM>
__device__ __forceinline__ void sts1e(float *base, int64_t e, float v)
{
    if constexpr (St<MM>::half) ((__bf16 *)base)[e] = (__bf16)v;
    else base[e] = v;
}
// The same through a WAVE-UNIFORM base pointer (kept in SGPRs: kernel arguments, tile coordinates) plus a 32-bit per-lane BYTE
// offset -- the SGPR-base + VGPR-offset form of global_load: no 64-bit vector add per load (the element-index forms above cost a
// v_lshl_add_u64 each; the PointNet data-gradient loop issued 16 of them per 12 MFMAs).  st_ptr advances a base by elements.
template <int MM>
__device__ __forceinline__ const float *st_ptr(const float *base, int64_t e)
{
    return (const float *)((const char *)base + e * St<MM>::bytes);
}
template <int MM>
__device__ __forceinline__ v4f lds4b(const float *sbase, unsigned boff)
{
    boff = fcn_opaque_v32(boff);
    if constexpr (St<MM>::half) {
        const bf16x4 h = *(gbf4p)((const char *)sbase + boff);
        return __builtin_convertvector(h, v4f);
    } else {
        return *(gv4fp)((const char *)sbase + boff);
    }
}
// what a stored value reads back as (BatchNorm sums are taken over the STORED values, so the statistics match the data)
template <int MM>
__device__ __forceinline__ float st_round(float v)
{
    if constexpr (St<MM>::half) return (float)(__bf16)v;
    else return v;
}
template <int MM>
__device__ __forceinline__ v4f st_round4(v4f v)
{
    if constexpr (St<MM>::half) return __builtin_convertvector(__builtin_convertvector(v, bf16x4), v4f);
    else return v;
}

// ------------------------------------------------------------------------------------------------
// "kb-major" operand images (PointNet forward / data-gradient GEMMs).  One u32x4 is EXACTLY what a lane feeds a 32x32x16 MFMA:
// the four packed dwords of 8 reduction-adjacent values of one row / column.  A 32-deep chunk of a T-row operand tile is two
// planes (hi parts, lo parts) of [4 k-blocks][LDR] u32x4, LDR = T + 2:
//   * fragment read  = ONE ds_read_b128 per operand part (256 B/clk; the k-major dword layout above needs four ds_read_b32 at
//     128 B/clk -- the PointNet GEMMs were bound by their LDS traffic, not by the matrix pipe);
//   * staging write  = ONE ds_write_b128 per part for the 8 values a thread loaded from 32 contiguous bytes of a row;
//   * the weight operand is PRE-ENCODED in this order in global memory once per step (pn_pack_*: fcn_pn_ws.wenc), so its
//     staging is a plain 16-byte copy, lane-linear in global memory and in LDS (no VALU, LDS-DMA ready);
//   * LDR = T + 2 keeps both conflict-free: 16 consecutive rows of one k-block cover all 64 banks for the b128 reads; the
//     8-lane groups of a b128 write (2 rows x 4 k-blocks, k-block stride 4 * LDR = 8 mod 32 dwords) cover 32 distinct banks.
// MM_F32 keeps the same geometry with raw floats: plane 0 = values 0..3 of the k-block, plane 1 = values 4..7.
#define KB_PAD 2
template <int T>
struct KbTile {
    static constexpr int LDR = T + KB_PAD;     // u32x4 per k-block row
    static constexpr int PLANE = 4 * LDR;      // u32x4 per plane of a 32-deep chunk
    static constexpr int U4 = 2 * PLANE;       // u32x4 per chunk
};

template <int MM>
__device__ __forceinline__ void enc8(const float (&x)[8], u32x4 &hi, u32x4 &lo)
{
    if constexpr (MM == MM_F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { hi[q] = __builtin_bit_cast(uint32_t, x[q]); lo[q] = __builtin_bit_cast(uint32_t, x[4 + q]); }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float h, l;
            enc2<MM>(x[2 * q], x[2 * q + 1], h, l);
            hi[q] = __builtin_bit_cast(uint32_t, h);
            lo[q] = __builtin_bit_cast(uint32_t, l);
        }
    }
}

// Stores the four reduction-adjacent values k = 4*kq .. 4*kq + 3 (chunk-local) of row r into a kb-major chunk image: one
// ds_write_b64 per part.  A thread that loaded ONE 16-byte piece of a row uses this -- the wave's load instruction then covers
// 8 rows x 128 contiguous bytes; having a thread load both halves of a k-block (32 B) instead makes every load instruction
// touch half lines of 16 rows (measured: the data-gradient kernels 15 % slower).
typedef float v2f_kb __attribute__((ext_vector_type(2)));
template <int MM, int LDR>
__device__ __forceinline__ void kb_store4(u32x4 *img, int r, int kq, float x0, float x1, float x2, float x3)
{
    const int kb = kq >> 1, half = kq & 1;
    if constexpr (MM == MM_F32) {
        const v4f v = {x0, x1, x2, x3};
        *(v4f *)(img + (half * 4 + kb) * LDR + r) = v;
    } else {
        float h0, l0, h1, l1;
        enc2<MM>(x0, x1, h0, l0);
        enc2<MM>(x2, x3, h1, l1);
        const v2f_kb h = {h0, h1}, l = {l0, l1};
        *(v2f_kb *)((float *)(img + kb * LDR + r) + 2 * half) = h;
        if constexpr (!mm_x1<MM>) *(v2f_kb *)((float *)(img + (4 + kb) * LDR + r) + 2 * half) = l;
    }
}

typedef const u32x4 __attribute__((address_space(1))) *gu4p;
__device__ __forceinline__ u32x4 ldgu4(const u32x4 *p) { return *(gu4p)p; }

// One 32-deep chunk: A is a KbTile<TA> image, B a KbTile<TB> image (LDRA / LDRB = their LDR).
template <int MM, int MT, int NT, int LDRA, int LDRB>
__device__ __forceinline__ void mma_chunk_kb(const u32x4 *A, const u32x4 *B, int arow0, int bcol0, f32x16 (&acc)[MT][NT])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, lh = lane >> 5;
    if constexpr (MM == MM_F32) {
        const float *Af = (const float *)A, *Bf = (const float *)B;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const int k = kk + lh, kb = k >> 3, j = k & 7;
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = Af[(((j >> 2) * 4 + kb) * LDRA + arow0 + i * 32 + l31) * 4 + (j & 3)];
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) b[jn] = Bf[(((j >> 2) * 4 + kb) * LDRB + bcol0 + jn * 32 + l31) * 4 + (j & 3)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
        }
    } else {
        constexpr bool X3 = !mm_x1<MM>;
        const u32x4 *ap = A + lh * LDRA + arow0 + l31;
        const u32x4 *bp = B + lh * LDRB + bcol0 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = ap[2 * ks * LDRA + i * 32];
                if constexpr (X3) al[i] = ap[(4 + 2 * ks) * LDRA + i * 32];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = bp[2 * ks * LDRB + j * 32];
                if constexpr (X3) bl[j] = bp[(4 + 2 * ks) * LDRB + j * 32];
            }
            if constexpr (X3) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(ah[i], bl[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(al[i], bh[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<MM>(ah[i], bh[j], acc[i][j]);
        }
    }
}

// Epilogue transposition patch of one wave: a 32 x 32 accumulator tile (lane = column, registers = rows) goes through a
// wave-private 32 x EP_LD float patch and comes back row-major, 4 consecutive columns per lane (idx = lane + 64 q: row idx >> 3,
// column quad idx & 7) -- so outputs leave as 16-byte stores (a quarter of the store instructions of one dword per lane).
// EP_LD = 32 (no padding) is the conflict-free choice for BOTH directions: the writes are 32 consecutive dwords of one row per
// half-wave, and the 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) touch four rows at column quads {0-3,4-7,4-7,0-3}
// -- 16-dword runs at 0, LD+16, 2LD+16, 3LD, which tile the 64 banks exactly when LD = 32 (with LD = 36 the second run wrapped
// onto the first: SQ_LDS_BANK_CONFLICT 9-16 % of the forward GEMMs' LDS cycles).
#define EP_LD 32
#define EP_FLOATS (32 * EP_LD)
__device__ __forceinline__ void ep_put(float *patch, const f32x16 &acc, int l31, int lh)
{
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) patch[acc_row_c(reg, lh) * EP_LD + l31] = acc[reg];
}
__device__ __forceinline__ v4f ep_get(const float *patch, int lane, int q)
{
    const int idx = lane + 64 * q;
    return *(const v4f *)(patch + (idx >> 3) * EP_LD + 4 * (idx & 7));
}

// host-side dispatch of a precision mode (fcn_pn_desc / fcn_cn_desc .precision) to the operand mode of a forward or a
// backward GEMM: f(std::integral_constant<int, MM>) -> int
#define FCN_MM_OF(prec, fwd)                                                                     \
    ((prec) == FCN_PREC_F32 ? MM_F32 : ((prec) == FCN_PREC_BF16 ? MM_BF16S : ((prec) == FCN_PREC_BF16_OPS ? MM_BF16X1 : ((fwd) ? MM_F16X3 : MM_BF16X3))))
#define FCN_MM_SWITCH(mm_, CALL)                                     \
    switch (mm_) {                                                   \
        case MM_F32: { constexpr int MM = MM_F32; CALL; } break;     \
        case MM_F16X3: { constexpr int MM = MM_F16X3; CALL; } break; \
        case MM_BF16X3: { constexpr int MM = MM_BF16X3; CALL; } break; \
        case MM_BF16X1: { constexpr int MM = MM_BF16X1; CALL; } break; \
        default: { constexpr int MM = MM_BF16S; CALL; } break;       \
    }

// XCD-aware tile order (cdna guide T1).  Workgroup ids are dealt to the 8 XCDs round-robin and every XCD has its own 4 MB
// L2: with the natural order the tiles that share operand rows land on eight different L2s (or on one L2 but a whole grid row
// apart in time) and each pulls the shared operand through the fabric again.  Workgroup `id` takes tile
// (id % 8) * ceil(n / 8) + id / 8: XCD k works through the CONTIGUOUS tile range k*per .. (k+1)*per-1 in order, so tiles
// that are adjacent in that order (the column tiles of one row tile) run back to back on one L2.  -1: no tile (grid padding).
// Speed only: nothing depends on the placement.
__device__ __forceinline__ int fcn_xcd_tile(int id, int n)
{
    const int per = (n + 7) >> 3;
    const int t = (id & 7) * per + (id >> 3);
    return (t < n && (id >> 3) < per) ? t : -1;
}

// Row (M index inside the 32x32 tile) held by accumulator register `reg` of this lane.
__device__ __forceinline__ int acc_row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }

// conv1 + BN1 + ReLU of one entry, folded: alpha = scale * W1 row, t = shift.  The SAME expression is
// used by the forward, the dgrad ReLU mask and the wgrad operand so the three agree bit-for-bit.
__device__ __forceinline__ float l1_pre(const float *al3, float t, float ux, float uy, float uz) {
    return fmaf(al3[0], ux, fmaf(al3[1], uy, fmaf(al3[2], uz, t)));
}
