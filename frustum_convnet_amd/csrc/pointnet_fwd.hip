// Forward of one PointNet scale in entry space (replaces models/det_base.py:75-101 + :134-157).
//
//   bn1_finalize : BN1 scale/shift from the weighted input moments (conv1 is linear in u)
//   fwd_gemm<0>  : a1 = relu(bn1(conv1(u))) computed on the fly (Cin = 3 -> VALU), y2 = a1 . W2^T on MFMA,
//                  per-channel weighted sum / sum-of-squares of y2 in the epilogue
//   bn_finalize  : BN scale/shift (+ running stats) from those sums
//   fwd_gemm<1>  : a2 = relu(bn2(y2)) applied while staging, y3 = a2 . W3^T, statistics epilogue
//   pool         : relu(bn3(y3)), (cnt>0) mask, max over each window's rows (+argmax), one-hot rows
//
// Training-mode BatchNorm needs whole-batch statistics between layers, hence one launch per layer;
// everything elementwise (gather, centre, BN apply, ReLU, mask, max, concat) is fused into the
// neighbouring GEMM's prologue/epilogue, so only the pre-BN conv outputs y2, y3 ever reach HBM.
#define FCN_TUNING_PNF
#include "gemm_tile.h"

#define LDT 129            // LDS leading dimension of a 128-wide k-major tile
#define MAXC 512           // largest supported reduction width (C1, C2)
// 128x256 (8-wave) tiles halve the staging work per output but leave ~1 workgroup per CU at B=32: the 128x128
// tiles balance better on 256 CUs (measured 124 vs 147 us on the 256->512 conv3), so they are the default.
#ifndef FCN_WIDE_TILES
#define FCN_WIDE_TILES 0
#endif
#ifndef FCN_C3_BIG
#define FCN_C3_BIG 1          // 128 x 128 tiles for the widest conv3 too (isolated on MI355X: 48.7 -> 44.1 us; 0: 64 x 128)
#endif
#ifndef FCN_FWD_EPI_DIRECT
#define FCN_FWD_EPI_DIRECT 0
#endif

// ------------------------------------------------------------------------------------------------
__global__ void bn1_finalize_kernel(const double *__restrict__ mom, const float *__restrict__ W1,
                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                    float *rmean, float *rvar, int64_t *nbt, int C, int training,
                                    float eps, float momentum, double M, float *__restrict__ bn)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        const double mx = mom[1] / M, my = mom[2] / M, mz = mom[3] / M;
        const double cxx = mom[4] / M - mx * mx, cxy = mom[5] / M - mx * my, cxz = mom[6] / M - mx * mz;
        const double cyy = mom[7] / M - my * my, cyz = mom[8] / M - my * mz, czz = mom[9] / M - mz * mz;
        const double w0 = W1[3 * c], w1 = W1[3 * c + 1], w2 = W1[3 * c + 2];
        mean = w0 * mx + w1 * my + w2 * mz;
        var = w0 * (cxx * w0 + cxy * w1 + cxz * w2) + w1 * (cxy * w0 + cyy * w1 + cyz * w2) +
              w2 * (cxz * w0 + cyz * w1 + czz * w2);
        if (var < 0.0) var = 0.0;
        if (rmean) {
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * (M / (M - 1.0)));
            if (c == 0 && nbt) nbt[0] += 1;
        }
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double s = (double)gamma[c] * rstd;
    bn[c] = (float)s;
    bn[C + c] = (float)((double)beta[c] - mean * s);
    bn[2 * C + c] = (float)mean;
    bn[3 * C + c] = (float)rstd;
}

__global__ void bn_finalize_kernel(const double *__restrict__ stat, int rep_stride, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float *rmean, float *rvar, int64_t *nbt,
                                   int C, int training, float eps, float momentum, double M,
                                   float *__restrict__ bn)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // (one fp64 division per thread, none per statistic, and no software sqrt: this launch sits between two layers of a
    // latency-bound chain)
    double mean, var;
    if (training) {
        const double invM = 1.0 / M;
        mean = fcn_rep_sum(stat + c, rep_stride) * invM;
        var = fcn_rep_sum(stat + C + c, rep_stride) * invM - mean * mean;
        if (var < 0.0) var = 0.0;
        if (rmean) {
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * (M / (M - 1.0)));
            if (c == 0 && nbt) nbt[0] += 1;
        }
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    const double rstd = fcn_rsqrt64(var + (double)eps);
    const double s = (double)gamma[c] * rstd;
    bn[c] = (float)s;
    bn[C + c] = (float)((double)beta[c] - mean * s);
    bn[2 * C + c] = (float)mean;
    bn[3 * C + c] = (float)rstd;
}

#include "pn_pack.h"

__global__ __launch_bounds__(256) void pn_pack_kernel(PackArgs a)
{
    pn_pack_range(a, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// (also called by fcn_pn_group_compact's launcher for all scales of a batch)
extern "C" int fcn_pn_pack_weights(const fcn_pn_desc *d, const fcn_pn_params *p, const fcn_pn_ws *ws, void *stream)
{
    if (!d || !p || !ws || !ws->wenc || !p->W[1] || !p->W[2]) return FCN_E_BADARG;
    if (d->C1 % 64 || d->C2 % 64 || d->C3 % 64) return FCN_E_BADARG;
    if (d->precision < 0 || d->precision > FCN_PREC_BF16_OPS || ((uintptr_t)ws->wenc & 15)) return FCN_E_BADARG;
    PackArgs a;
    a.W2 = p->W[1]; a.W3 = p->W[2]; a.wenc = ws->wenc; a.C1 = d->C1; a.C2 = d->C2; a.C3 = d->C3; a.precision = d->precision;
    const int items = 2 * (d->C2 * d->C1 + d->C3 * d->C2) / 8;
    hipLaunchKernelGGL(pn_pack_kernel, dim3((items + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

struct PackAll {
    PackArgs s[8];
};
__global__ __launch_bounds__(256) void pn_pack_all_kernel(PackAll a)
{
    // the scale is workgroup-uniform: static indices only (a dynamically indexed kernel argument goes through scratch)
    PackArgs S = a.s[0];
#pragma unroll
    for (int q = 1; q < 8; ++q)
        if ((int)blockIdx.y == q) S = a.s[q];
    pn_pack_range(S, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

extern "C" int fcn_pn_pack_weights_all(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p,
                                       const fcn_pn_ws *const *ws, void *stream)
{
    if (nscale < 1 || nscale > 8 || !d || !p || !ws) return FCN_E_BADARG;
    PackAll a;
    int maxitems = 0;
    for (int s = 0; s < 8; ++s) {
        const int q = s < nscale ? s : 0;
        if (!d[q] || !p[q] || !ws[q] || !ws[q]->wenc || !p[q]->W[1] || !p[q]->W[2]) return FCN_E_BADARG;
        if (d[q]->C1 % 64 || d[q]->C2 % 64 || d[q]->C3 % 64) return FCN_E_BADARG;
        if (d[q]->precision < 0 || d[q]->precision > FCN_PREC_BF16_OPS || ((uintptr_t)ws[q]->wenc & 15)) return FCN_E_BADARG;
        PackArgs &S = a.s[s];
        S.W2 = p[q]->W[1]; S.W3 = p[q]->W[2]; S.wenc = ws[q]->wenc; S.C1 = d[q]->C1; S.C2 = d[q]->C2; S.C3 = d[q]->C3;
        S.precision = d[q]->precision;
        const int items = 2 * (S.C2 * S.C1 + S.C3 * S.C2) / 8;
        if (s < nscale && items > maxitems) maxitems = items;
    }
    // a few items per thread: the widest scale sets the grid, the narrow ones finish early
    hipLaunchKernelGGL(pn_pack_all_kernel, dim3((maxitems + 1023) / 1024, nscale), dim3(256), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------

struct FwdArgs {
    const float4 *ent;     // (B,cap) rows (ux,uy,uz,w)
    const int32_t *woff;   // (B,L+1)
    const int32_t *tiles;  // live-tile list
    const float *aprev;    // MODE 1: (B,cap,CIN) pre-BN output of the previous conv
    const float *bn_in;    // scale[CIN], shift[CIN] of the BN in front of this conv
    const float *W1;       // MODE 0: (CIN,3)
    const u32x4 *Wenc;     // forward image of the conv weight (pn_pack_*): [CIN/32][2][4][COUT]
    int32_t *flags;        // sticky numeric flags (fcn_pn_ws.flags) or nullptr
    float *y;              // (B,cap,COUT)
    double *stat;          // sum[COUT], sumsq[COUT] or nullptr (eval)
    int L, cap, CIN, COUT, tps;
    // MODE 1 with gamma_in set: the BN in front of this conv is FINALISED BY ITS CONSUMER -- every workgroup derives scale /
    // shift in its prologue from the batch sums (training) or the running statistics (eval) instead of waiting for a
    // one-workgroup launch between the two GEMMs; workgroup 0 also publishes scale, shift, mean, rstd (the backward reads
    // them) and updates the running statistics.  gamma_in null: bn_in holds finished scale / shift.
    const double *stat_in;      // replica 0 of sum[CIN], sumsq[CIN], or nullptr (eval)
    int rep_stride;             // doubles between the replica blocks of stat / stat_in
    const float *gamma_in, *beta_in;
    float *rmean_in, *rvar_in;
    int64_t *nbt_in;
    float *bn_pub;              // 4 x CIN
    double M;
    float eps, momentum;
    // POOL (conv3 with the max-pool folded into its epilogue): window of each row, the keys (B, L, COUT) -- zero between
    // launches --, gamma of the BN behind this conv (its sign orients the max)
    const int32_t *ewin;
    unsigned long long *pkey;
    const float *gamma_out;
};

// (s1, s2: the channel's batch sums when the caller fetched them already -- `have` --, else they are read here)
__device__ __forceinline__ void fwd_bn_in(const FwdArgs &a, int i, bool pub, float &fs, float &ft, bool have = false, double s1 = 0.0,
                                          double s2 = 0.0)
{
    const int C = a.CIN;
    double mean, var;
    if (a.stat_in) {
        const double invM = 1.0 / a.M;
        if (!have) {
            s1 = fcn_rep_sum(a.stat_in + i, a.rep_stride);
            s2 = fcn_rep_sum(a.stat_in + C + i, a.rep_stride);
        }
        mean = s1 * invM;
        var = s2 * invM - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = a.rmean_in[i];
        var = a.rvar_in[i];
    }
    const double rstd = fcn_rsqrt64(var + (double)a.eps);
    const double sc = (double)a.gamma_in[i] * rstd;
    fs = (float)sc;
    ft = (float)((double)a.beta_in[i] - mean * sc);
    if (pub) {
        a.bn_pub[i] = fs; a.bn_pub[C + i] = ft; a.bn_pub[2 * C + i] = (float)mean; a.bn_pub[3 * C + i] = (float)rstd;
        if (a.stat_in && a.rmean_in) {
            a.rmean_in[i] = (float)((1.0 - a.momentum) * a.rmean_in[i] + a.momentum * mean);
            a.rvar_in[i] = (float)((1.0 - a.momentum) * a.rvar_in[i] + a.momentum * var * (a.M / (a.M - 1.0)));
            if (i == 0 && a.nbt_in) a.nbt_in[0] += 1;
        }
    }
}

// MODE 0: operand rows are conv1+BN1+ReLU of the entries (computed here); MODE 1: BN+ReLU of aprev.
// Workgroup = 2 x WN waves; tile = 128 rows x (32*NT*WN) output channels.  WN = 4 (512 threads) shares one
// staged A tile between 256 output columns, halving the BN+ReLU staging work per output element.
// MT = 1: 64-row tiles, two workgroups per listed 128-row tile (scale 4's conv3: 1140 big tiles are 1.5 waves of the 768
// resident slots -- two rounds; 2280 half tiles on 1024 slots are 2.2 half rounds).
//
// POOL = 1: the window max of relu(bn(y)) is taken HERE, over the accumulators, as (value, row) keys (fcn_pool_key): lanes own
// columns and walk their rows in order, a segment per window; the segments of a workgroup meet in an LDS table [window slot]
// [column] (ds_max_u64), and the table leaves as one 8-byte store per (window, column) when the window lies inside the tile, one
// device-scope atomic max when it continues in a neighbouring tile (two windows per tile at most).  pool_keys_kernel turns the
// keys into features once the batch statistics are complete -- the pooling pass that re-read all of y3 is gone.
template <int MM, int MODE, int NT, int WN, int MT = 2, int POOL = 0>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 4 : 3, 4))) void fwd_gemm_kernel(FwdArgs a)
{
    constexpr int NTHR = 128 * WN;
    constexpr int TM = 64 * MT;               // rows of the tile
    constexpr int SUB = 128 / TM;             // workgroups per listed 128-row tile
    constexpr int TN = 32 * NT * WN;
    constexpr int LDRA = KbTile<TM>::LDR, LDRB = KbTile<TN>::LDR;
    constexpr int NA4 = TM * 8 / NTHR;        // 16-byte pieces of A per thread per chunk (MODE 1)
    constexpr int NB = TN * 8 / NTHR;         // u32x4 of the encoded weight per thread per chunk
    constexpr int KBT = 4 * TM / NTHR;        // MODE 0: k-blocks per thread per chunk
    constexpr int OPU4 = KbTile<TM>::U4 + KbTile<TN>::U4, EPU4 = (NTHR / 64) * EP_FLOATS / 4;
    constexpr int LDSU4 = OPU4 > EPU4 ? OPU4 : EPU4;      // operand images; the waves' epilogue patches alias them
    __shared__ u32x4 lds4[LDSU4];
    __shared__ __attribute__((aligned(16))) float tS[MAXC];
    __shared__ __attribute__((aligned(16))) float sS[MODE == 0 ? 3 * MAXC : MAXC];   // MODE 0: alpha[CIN][3]
    __shared__ float wS[TM];
    __shared__ int winS[POOL ? TM : 1], insS[POOL ? 64 : 1];
    u32x4 *Ab = lds4, *Bb = lds4 + KbTile<TM>::U4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // live (row tile, column tile) pairs in XCD order, column tiles fastest: the workgroups that stage the same rows run
    // back to back on one L2
    const int ny = a.COUT / TN;
    if constexpr (MODE == 1) {
        if (blockIdx.x == 0 && a.gamma_in) {     // workgroup 0 always exists (the grid is padded) whatever the live-tile list holds
            for (int i = tid; i < a.CIN; i += NTHR) {
                float fs, ft;
                fwd_bn_in(a, i, true, fs, ft);
            }
        }
    }
    // MODE 1, training: the batch sums of this thread's input channel are requested FIRST -- 2 x FCN_STAT_REP loads whose latency
    // then runs beside the three dependent round trips of the tile lookup below instead of after them (tools/pn_probe.py fwd: the
    // prologue was 22 % of a conv3 workgroup's cycles on the widest scale and 40 % on the narrow ones)
    double q1[FCN_STAT_REP], q2[FCN_STAT_REP];
    const bool early = FCN_EARLY_STATS && MODE == 1 && a.gamma_in && a.stat_in && tid < a.CIN;
    if constexpr (MODE == 1) {
        if (early) {
#pragma unroll
            for (int r = 0; r < FCN_STAT_REP; ++r) {
                q1[r] = a.stat_in[(int64_t)r * a.rep_stride + tid];
                q2[r] = a.stat_in[(int64_t)r * a.rep_stride + a.CIN + tid];
            }
        }
        FCN_LOAD_FENCE();         // (without it the loads are sunk below the early returns of the tile lookup, to where their values are used)
    }
    const int xt = fcn_xcd_tile(blockIdx.x, SUB * a.tiles[0] * ny);
    if (xt < 0) return;
    const int bxi = xt / ny, byi = xt % ny;
    const int lt = bxi / SUB, sub = bxi % SUB;
    const int code = a.tiles[4 + lt];
    const int b = code / a.tps, t = code % a.tps;
    const int nent = a.woff[(int64_t)b * (a.L + 1) + a.L];
    const int row0 = t * 128 + sub * TM;
    const int nvalid = min(TM, nent - row0);
    if (nvalid <= 0) return;
    const int64_t grow0 = (int64_t)b * a.cap + row0;
    const int n0 = byi * TN;
    const int CIN = a.CIN, COUT = a.COUT;

    PNF_DECL;
    u32x4 rw[NB];
    v4f ra[MODE == 1 ? NA4 : 1];
    const u32x4 *wsrc = a.Wenc + n0 + (tid % TN) + (int64_t)(tid / TN) * COUT;      // item f = tid + NTHR*i: column f % TN, (plane, k-block) f / TN
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        if constexpr (MODE == 1) {
            if (!((FCN_X & 1) && c > 0))
#pragma unroll
            for (int i = 0; i < NA4; ++i) {
                const int f = tid + NTHR * i;
                const int r = f >> 3, kq = f & 7;
                // unconditional, from a clamped row (a "load or zero" select makes the compiler wait for the load at
                // once); rows past nvalid are zeroed when the registers go to LDS.  32-bit element offsets
                // (launch_fwd_gemm checks B * cap * CIN < 2^31).
                ra[i] = lds4e<MM>(a.aprev, ((int)grow0 + min(r, nvalid - 1)) * CIN + c * KC + 4 * kq);
            }
        }
        if (!((FCN_X & 2) && c > 0))
#pragma unroll
        for (int i = 0; i < NB; ++i) rw[i] = ldgu4(wsrc + ((int64_t)c * 8 + i * (NTHR / TN)) * COUT);
    };
    load_chunk(0);             // in flight across the prologue below

    for (int i = tid; i < CIN; i += NTHR) {
        float s, t;
        if (MODE == 1 && a.gamma_in) {
            if (early && i == tid) {
                double s1 = q1[0], s2 = q2[0];
#pragma unroll
                for (int r = 1; r < FCN_STAT_REP; ++r) { s1 += q1[r]; s2 += q2[r]; }      // replica order, like fcn_rep_sum
                fwd_bn_in(a, i, false, s, t, true, s1, s2);
            } else {
                fwd_bn_in(a, i, false, s, t);
            }
        } else {
            s = a.bn_in[i];
            t = a.bn_in[CIN + i];
        }
        tS[i] = t;
        if constexpr (MODE == 0) {
            sS[3 * i] = s * a.W1[3 * i];
            sS[3 * i + 1] = s * a.W1[3 * i + 1];
            sS[3 * i + 2] = s * a.W1[3 * i + 2];
        } else {
            sS[i] = s;
        }
    }
    if (tid < TM) wS[tid] = (tid < nvalid) ? a.ent[grow0 + tid].w : 0.f;
    // POOL: window of each row, and -- kept in registers until the epilogue -- whether that window lies inside this tile (its
    // row range from woff: two dependent loads that have the whole K loop to arrive)
    int pw = -1, plo = 0, phi = 0;
    if constexpr (POOL) {
        if (tid < TM) {
            if (tid < nvalid) {
                pw = a.ewin[grow0 + tid];
                plo = a.woff[(int64_t)b * (a.L + 1) + pw];
                phi = a.woff[(int64_t)b * (a.L + 1) + pw + 1];
            }
            winS[tid] = pw;
        }
    }
    float ux = 0.f, uy = 0.f, uz = 0.f;
    const int r0 = tid % TM;
    const bool r0valid = r0 < nvalid;
    const float r0keep = fcn_keep(r0valid);
    if (MODE == 0 && r0valid) {
        const float4 e = a.ent[grow0 + r0];
        ux = e.x; uy = e.y; uz = e.z;
    }
    __syncthreads();

    f32x16 acc[MT][NT];
    acc_zero<MT, NT>(acc);
    const int nchunk = CIN / KC;

    // ---- registers -> LDS (kb-major images of chunk c at Aq / Bq), applying the input BN + ReLU
    auto stage_chunk = [&](int c, u32x4 *Aq, u32x4 *Bq) __attribute__((always_inline)) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NA4; ++i) {
                const int f = tid + NTHR * i;
                const int r = f >> 3, kq = f & 7;
                const float kp = fcn_keep(r < nvalid);          // (loop-invariant) ReLU + row mask as ONE v_med3_f32 per element
                const v4f s4 = *(const v4f *)(sS + c * KC + 4 * kq), t4 = *(const v4f *)(tS + c * KC + 4 * kq);
                const float z0 = fcn_relu_keep(fmaf(s4.x, ra[i].x, t4.x), kp), z1 = fcn_relu_keep(fmaf(s4.y, ra[i].y, t4.y), kp);
                const float z2 = fcn_relu_keep(fmaf(s4.z, ra[i].z, t4.z), kp), z3 = fcn_relu_keep(fmaf(s4.w, ra[i].w, t4.w), kp);
                kb_store4<MM_ENC_A, LDRA>(Aq, r, kq, z0, z1, z2, z3);
            }
        } else {
            const int part = tid / TM;
#pragma unroll
            for (int q = 0; q < KBT; ++q) {
                const int kb = part * KBT + q;
                float z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = c * KC + 8 * kb + j;
                    const float zz = l1_pre(&sS[3 * kk], tS[kk], ux, uy, uz);
                    z[j] = fcn_relu_keep(zz, r0keep);
                }
                u32x4 hi, lo;
                enc8<MM_ENC_A>(z, hi, lo);
                Aq[kb * LDRA + r0] = hi;
                Aq[(4 + kb) * LDRA + r0] = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + NTHR * i;
            Bq[(f / TN) * LDRB + (f % TN)] = rw[i];       // (plane, k-block) row f / TN of the image
        }
    };

    PNF_ADD(0);                                   // 0: prologue (BN of the input, row weights, first loads issued)
    for (int c = 0; c < nchunk; ++c) {
        if (!((FCN_X & 4) && c > 0)) stage_chunk(c, Ab, Bb);
        PNF_ADD(2);                               // 2: wait for the loads + operand transform + LDS stores
        __syncthreads();
        PNF_ADD(3);                               // 3: barriers
        if (c + 1 < nchunk) load_chunk(c + 1);
        PNF_ADD(1);                               // 1: issue of the global loads
        if (!(FCN_X & 8)) mma_chunk_kb<MM, MT, NT, LDRA, LDRB>(Ab, Bb, wm * 32 * MT, wn * 32 * NT, acc);
        PNF_ADD(4);                               // 4: LDS operand reads + MFMAs
        __syncthreads();
        PNF_ADD(3);
    }

    // ---- epilogue: y out as 16-byte stores through the wave's transposition patch, per-channel weighted statistics
    if (FCN_X & 16) { if (acc[0][0][0] == 123.456f) a.y[0] = 0.f; return; }
    bool bad = false;
    // (eval mode with the pooling in this epilogue: nothing reads y afterwards -- the keys carry the pooled values)
    const bool store_y = !POOL || a.stat != nullptr;
#if FCN_FWD_EPI_DIRECT      // (tuning builds: one dword per lane straight from the accumulator layout -- the round-2 epilogue)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wm * 32 * MT + mt * 32 + acc_row(reg, lh);
                const int col = n0 + wn * 32 * NT + nt * 32 + l31;
                const float v = acc[mt][nt][reg];
                if constexpr (MM == MM_F16X3) bad |= !(fabsf(v) < 3.0e38f);
                if (row < nvalid && store_y) sts1e<MM>(a.y, (grow0 + row) * COUT + col, v);
            }
#else
    float *patch = (float *)lds4 + wave * EP_FLOATS;          // (the operand buffers are free after the last barrier)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ep_put(patch, acc[mt][nt], l31, lh);
            __builtin_amdgcn_wave_barrier();
            const int rbase = wm * 32 * MT + mt * 32, cbase = n0 + wn * 32 * NT + nt * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane + 64 * q, row = rbase + (idx >> 3);
                const v4f v = ep_get(patch, lane, q);
                if constexpr (MM == MM_F16X3) bad |= !(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) < 3.0e38f);
                if (row < nvalid && store_y && !(FCN_X & 32)) sts4e<MM>(a.y, (grow0 + row) * COUT + cbase + 4 * (idx & 7), v);
            }
            __builtin_amdgcn_wave_barrier();
        }
#endif
    PNF_ADD(5);                                   // 5: y stores (the rest up to the total: pooling keys, statistics)
    if constexpr (MM == MM_F16X3) {
        // fp16 operand parts overflow at |x| >= 65504 (inf - inf = NaN in the products, which a later ReLU would turn into a
        // silent zero): a non-finite output raises the sticky flag of the workspace
        if (a.flags && __ballot(bad) != 0ull && lane == 0) atomicOr(a.flags, FCN_FLAG_NONFINITE);
    }
    if constexpr (POOL) {
        constexpr int NSLOT = (LDSU4 * 2 / TN) < 64 ? (LDSU4 * 2 / TN) : 64;      // window slots of the LDS table (windows past it: direct atomics)
        unsigned long long *tab = (unsigned long long *)lds4;
        unsigned long long *pk = a.pkey + (int64_t)b * a.L * COUT + n0;
        const int win0 = winS[0];
        const int ns = min(winS[nvalid - 1] - win0 + 1, NSLOT);
        __syncthreads();                                      // every wave is done with its patch (the table aliases them)
        for (int i = tid; i < ns * TN / 2; i += NTHR) lds4[i] = u32x4{0u, 0u, 0u, 0u};
        if (pw >= 0 && pw - win0 < NSLOT) insS[pw - win0] = (plo >= row0 && phi <= row0 + nvalid) ? 1 : 0;
        __syncthreads();
        {
            // a lane's segment = its rows of one window, all NT columns of the lane at once: (oriented value, row) of the best so
            // far, a strict > keeps the earlier row; the 64-bit keys are only built when the segment ends
            float g[NT], bval[NT];
            int brow[NT], cur = -1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { g[nt] = a.gamma_out[n0 + wn * 32 * NT + nt * 32 + l31]; bval[nt] = 0.f; brow[nt] = 0; }
            auto flush = [&]() __attribute__((always_inline)) {
                if (cur < 0) return;
                const int slot = cur - win0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = wn * 32 * NT + nt * 32 + l31;
                    const unsigned long long best = fcn_pool_key(bval[nt], row0 + brow[nt]);
                    // (different scopes also keep the compiler from merging the two into one FLAT atomic on a selected address)
                    if (slot < NSLOT) __hip_atomic_fetch_max(&tab[slot * TN + col], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_max(&pk[(int64_t)cur * COUT + col], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {          // (rows ascend with mt, reg: windows are runs of rows)
                    const int row = wm * 32 * MT + mt * 32 + acc_row(reg, lh);
                    const int w = winS[row];
                    const bool fresh = w != cur;
                    if (fresh) {
                        flush();
                        cur = w;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float v = fcn_pool_orient(st_round<MM>(acc[mt][nt][reg]), g[nt]);
                        if (fresh || v > bval[nt]) {
                            bval[nt] = v;
                            brow[nt] = row;
                        }
                    }
                }
            flush();
        }
        __syncthreads();
        for (int i = tid; i < ns * TN; i += NTHR) {
            const int slot = i / TN, col = i % TN;
            const unsigned long long key = tab[i];
            unsigned long long *dst = &pk[(int64_t)(win0 + slot) * COUT + col];
            // a window inside this tile has no rows anywhere else: plain store (the keys are zero between launches)
            if (insS[slot]) *dst = key;
            else __hip_atomic_fetch_max(dst, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.stat) {
        __syncthreads();                                      // every wave is done with its patch (red aliases them)
        float *red = (float *)lds4;   // [wn][nt*32+l31][2] written by the wm==1 waves
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float w = wS[wm * 32 * MT + mt * 32 + acc_row(reg, lh)];
                    const float v = st_round<MM>(acc[mt][nt][reg]);       // (the sums are over the values as stored)
                    s1 = fmaf(w, v, s1);
                    s2 = fmaf(w * v, v, s2);
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (wm == 1 && lh == 0) {
                red[((wn * NT + nt) * 32 + l31) * 2] = s1;
                red[((wn * NT + nt) * 32 + l31) * 2 + 1] = s2;
            }
            acc[0][nt][0] = s1;   // keep for the wm == 0 waves
            acc[0][nt][1] = s2;
        }
        __syncthreads();
        if (wm == 0 && lh == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + wn * 32 * NT + nt * 32 + l31;
                const double s1 = (double)acc[0][nt][0] + (double)red[((wn * NT + nt) * 32 + l31) * 2];
                const double s2 = (double)acc[0][nt][1] + (double)red[((wn * NT + nt) * 32 + l31) * 2 + 1];
                double *sr = a.stat + (int64_t)fcn_rep_id() * a.rep_stride;
                if (!(FCN_X & 64)) {
                    atomic_add_f64(&sr[col], s1);
                    atomic_add_f64(&sr[COUT + col], s2);
                } else if (s1 == 123.456) sr[col] = s2;
            }
        }
    }
    PNF_FLUSH(((unsigned long long)(MODE + 2) << 48) | ((unsigned long long)CIN << 32) | ((unsigned long long)COUT << 16) |
              (unsigned long long)nvalid);
}

// ------------------------------------------------------------------------------------------------
// Pool: one wave = one window x 64 channels (lane = channel: every row read is a coalesced 256-B run);
// a workgroup covers 16 consecutive windows so the (B, C, L) output goes out as 64-B runs through LDS.
#define PW 4            // windows per workgroup: ONE PER WAVE (the row walk of a window is a serial latency chain)

template <int S16>
__global__ __launch_bounds__(GT) void pool_kernel(
    const float *__restrict__ y3, const float *__restrict__ bn3, const int32_t *__restrict__ woff,
    const int32_t *__restrict__ cnt, const float *__restrict__ one_hot, float *__restrict__ feat,
    int32_t *__restrict__ amax, int L, int cap, int C3, int nvec, int nlc, double *__restrict__ zero_ptr, int zero_n)
{
    __shared__ float outS[64 * (PW + 1)];
    // the BN-backward sum buffer of this scale is zeroed by the last forward kernel: no memset node heading the backward
    if (zero_ptr && blockIdx.x == 0 && blockIdx.z == 0)
        for (int i = blockIdx.y * GT + threadIdx.x; i < zero_n; i += gridDim.y * GT) zero_ptr[i] = 0.0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, l0 = blockIdx.x * PW, c0 = blockIdx.y * 64;
    const int c = c0 + lane;
    const float s = bn3[c], t = bn3[C3 + c];
    const int32_t *wo = woff + (int64_t)b * (L + 1);
    {
        const int wl = wave;
        const int l = l0 + wl;
        float best = 0.f;
        int arg = -1;
        if (l < L && cnt[(int64_t)b * L + l] > 0) {
            const int o0 = wo[l], o1 = wo[l + 1];
            constexpr int SM = S16 ? MM_BF16S : MM_F32;         // storage of y3: bf16 in the bf16 throughput mode
            const int64_t yo = ((int64_t)b * cap + o0) * C3 + c;
            int r = o0;
            // 8 independent row loads in flight; compared in row order (first maximum wins, like torch.max)
            for (; r + 8 <= o1; r += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lds1e<SM>(y3, yo + (int64_t)(r - o0 + j) * C3);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = fmaf(s, v[j], t);
                    if (u > best) { best = u; arg = r + j; }
                }
            }
            for (; r < o1; ++r) {
                const float v = fmaf(s, lds1e<SM>(y3, yo + (int64_t)(r - o0) * C3), t);
                if (v > best) { best = v; arg = r; }
            }
        }
        outS[lane * (PW + 1) + wl] = best;
        if (l < L && amax) amax[((int64_t)b * L + l) * C3 + c] = arg;
        if (nlc && l < L) feat[((int64_t)b * L + l) * C3 + c] = best;      // position-major: lanes = channels
    }
    if (nlc) return;
    __syncthreads();
    const int CT = C3 + nvec;
    for (int f = tid; f < 64 * PW; f += GT) {
        const int cc = f / PW, wl = f % PW, l = l0 + wl;
        if (l < L) feat[((int64_t)b * CT + c0 + cc) * L + l] = outS[cc * (PW + 1) + wl];
    }
    if (blockIdx.y == 0 && tid < nvec * PW) {
        const int v = tid / PW, wl = tid % PW, l = l0 + wl;
        if (l < L) feat[((int64_t)b * CT + C3 + v) * L + l] = one_hot[(int64_t)b * nvec + v];
    }
}

// Position-major pooling (the fused path: feat (B, L, C3) feeds the ConvFeatNet GEMMs directly).  A wave covers ALL C3
// channels of a row: lane = C3/64 consecutive channels, so a row is one fully coalesced C3*4-byte read per wave (the
// 64-channel slices of pool_kernel kept only 256 B per row in flight: 2.1 TB/s on the widest scale) and UN rows are in flight
// at once.  WPW waves share one window, taking its UN-row batches round-robin: the row walk of a window is a serial chain of
// memory round trips, and a full window (nsample rows: 32 batches of 4 on the widest scale) set the kernel's duration --
// 58 us for 74 MB -- while most waves had finished long before.  The partial (max, arg-max) pairs meet in LDS; ties go to
// the earlier row, like torch.max.
typedef float v2f __attribute__((ext_vector_type(2)));
// BN3 as its consumer sees it: every pooling workgroup derives scale / shift of all C3 channels from the (replicated) batch sums in
// its prologue -- the one-workgroup bn_finalize launch between conv3 and the pooling is gone from every scale's chain --, and
// workgroup (0,0,0) publishes scale, shift, mean, rstd for the backward and updates the running statistics.
struct PoolBn {
    const double *stat;        // replica 0 of sum[C3], sumsq[C3] (training), or nullptr: running statistics
    int rep_stride;
    const float *gamma, *beta;
    float *rmean, *rvar;
    int64_t *nbt;
    float *bn;                 // 4 x C3 published
    double M;
    float eps, momentum;
};
template <int VEC, int WPW, int S16>
__global__ __launch_bounds__(GT) void pool_nlc_kernel(
    const float *__restrict__ y3, PoolBn q, const int32_t *__restrict__ woff,
    const int32_t *__restrict__ cnt, float *__restrict__ feat, int32_t *__restrict__ amax, int L, int cap, int C3,
    double *__restrict__ zero_ptr, int zero_n)
{
    constexpr int UN = VEC >= 8 ? 4 : 8;          // rows in flight per lane
    constexpr int WIN = PW / WPW;                 // windows per workgroup
    __shared__ float bS[WPW > 1 ? GT * VEC : 1];
    __shared__ int aS[WPW > 1 ? GT * VEC : 1];
    __shared__ float scS[MAXC], shS[MAXC];
    // the BN-backward sum buffer of this scale is zeroed by the last forward kernel: no memset node heading the backward
    if (zero_ptr && blockIdx.z == 0)
        for (int i = blockIdx.x * GT + threadIdx.x; i < zero_n; i += gridDim.x * GT) zero_ptr[i] = 0.0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, l = blockIdx.x * WIN + wave / WPW, part = wave % WPW;
    const bool live = l < L;
    const int c = lane * VEC;
    {
        const bool pub = blockIdx.x == 0 && blockIdx.z == 0;
        for (int ch = tid; ch < C3; ch += GT) {
            double mean, var;
            if (q.stat) {
                const double invM = 1.0 / q.M;
                mean = fcn_rep_sum(q.stat + ch, q.rep_stride) * invM;
                var = fcn_rep_sum(q.stat + C3 + ch, q.rep_stride) * invM - mean * mean;
                if (var < 0.0) var = 0.0;
            } else {
                mean = q.rmean[ch];
                var = q.rvar[ch];
            }
            const double rstd = fcn_rsqrt64(var + (double)q.eps);
            const double sc = (double)q.gamma[ch] * rstd;
            const float fs = (float)sc, ft = (float)((double)q.beta[ch] - mean * sc);
            scS[ch] = fs;
            shS[ch] = ft;
            if (pub) {
                q.bn[ch] = fs; q.bn[C3 + ch] = ft; q.bn[2 * C3 + ch] = (float)mean; q.bn[3 * C3 + ch] = (float)rstd;
                if (q.stat && q.rmean) {
                    q.rmean[ch] = (float)((1.0 - q.momentum) * q.rmean[ch] + q.momentum * mean);
                    q.rvar[ch] = (float)((1.0 - q.momentum) * q.rvar[ch] + q.momentum * var * (q.M / (q.M - 1.0)));
                    if (ch == 0 && q.nbt) q.nbt[0] += 1;
                }
            }
        }
        __syncthreads();
    }
    float s[VEC], t[VEC], best[VEC];
    int arg[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s[v] = scS[c + v]; t[v] = shS[c + v]; best[v] = 0.f; arg[v] = -1; }
    if (live && cnt[(int64_t)b * L + l] > 0) {
        const int32_t *wo = woff + (int64_t)b * (L + 1);
        const int o0 = wo[l], o1 = wo[l + 1];
        constexpr int SM = S16 ? MM_BF16S : MM_F32;             // storage of y3: bf16 in the bf16 throughput mode
        const int64_t ybase = (int64_t)b * cap * C3 + c;
        // batches part, part + WPW, ...: rows past the window's end are loaded from its last row (unconditional loads) and
        // skipped in the comparison; rows are compared in row order, the first maximum wins
        for (int r = o0 + part * UN; r < o1; r += WPW * UN) {
            float v[UN][VEC];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int64_t q = ybase + (int64_t)min(r + j, o1 - 1) * C3;
                if constexpr (VEC == 2) {
                    if constexpr (S16) {
                        const bf16x2 x = *(const bf16x2 __attribute__((address_space(1))) *)((const __bf16 *)y3 + q);
                        v[j][0] = (float)x[0]; v[j][1] = (float)x[1];
                    } else {
                        const v2f x = *(const v2f __attribute__((address_space(1))) *)(y3 + q);
                        v[j][0] = x.x; v[j][1] = x.y;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < VEC / 4; ++h) {
                        const v4f x = lds4e<SM>(y3, q + 4 * h);
                        v[j][4 * h] = x.x; v[j][4 * h + 1] = x.y; v[j][4 * h + 2] = x.z; v[j][4 * h + 3] = x.w;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const bool ok = r + j < o1;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float u = fmaf(s[e], v[j][e], t[e]);
                    if (ok && u > best[e]) { best[e] = u; arg[e] = r + j; }
                }
            }
        }
    }
    if constexpr (WPW > 1) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { bS[e * GT + tid] = best[e]; aS[e * GT + tid] = arg[e]; }
        __syncthreads();
        if (part != 0) return;
#pragma unroll
        for (int q = 1; q < WPW; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float b2 = bS[e * GT + tid + 64 * q];
                const int a2 = aS[e * GT + tid + 64 * q];
                // (arg >= 0 implies best > 0; equal maxima: the earlier row)
                if (b2 > best[e] || (b2 == best[e] && a2 >= 0 && (arg[e] < 0 || a2 < arg[e]))) { best[e] = b2; arg[e] = a2; }
            }
    }
    if (!live) return;
    float *fo = feat + ((int64_t)b * L + l) * C3 + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) fo[e] = best[e];
    if (amax) {
        int32_t *ao = amax + ((int64_t)b * L + l) * C3 + c;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ao[e] = arg[e];
    }
}

// Pooling from the keys conv3's epilogue left (fwd_gemm_kernel, POOL): feat (B, L, C3) = relu(bn3(y of the winning row)), amax =
// that row (-1: nothing positive, no gradient), and the keys go back to zero for the next launch.  A thread owns two adjacent
// channels (16-byte key loads) and derives their BN3 scale / shift itself from the batch sums; the threads of workgroup (0, 0)
// that sit on the first window publish scale, shift, mean, rstd and update the running statistics.  WPB windows per workgroup.
__global__ __launch_bounds__(GT) void pool_keys_kernel(unsigned long long *__restrict__ pkey, PoolBn q,
                                                       const int32_t *__restrict__ cnt, float *__restrict__ feat,
                                                       int32_t *__restrict__ amax, float *__restrict__ ywin, int L, int C3, int WPB,
                                                       double *__restrict__ zero_ptr, int zero_n,
                                                       const void *__restrict__ y3, int cap, int s16)
{
    if (zero_ptr && blockIdx.y == 0)
        for (int i = blockIdx.x * GT + threadIdx.x; i < zero_n; i += gridDim.x * GT) zero_ptr[i] = 0.0;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int c = (2 * tid) % C3, sub = (2 * tid) / C3, nsub = 2 * GT / C3;       // C3 <= 2 * GT
    float g[2], s[2], t[2];
    const bool pub = blockIdx.x == 0 && blockIdx.y == 0 && sub == 0;
    {
        // the two channels' sums as 16-byte loads, every replica requested before the first is used (one memory round trip)
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 mean, var;
        if (q.stat) {
            d2 s1[FCN_STAT_REP], s2[FCN_STAT_REP];
#pragma unroll
            for (int r = 0; r < FCN_STAT_REP; ++r) {
                s1[r] = *(const d2 *)(q.stat + (int64_t)r * q.rep_stride + c);
                s2[r] = *(const d2 *)(q.stat + (int64_t)r * q.rep_stride + C3 + c);
            }
            d2 a1 = s1[0], a2 = s2[0];
#pragma unroll
            for (int r = 1; r < FCN_STAT_REP; ++r) { a1 += s1[r]; a2 += s2[r]; }       // replica order, like fcn_rep_sum
            const double invM = 1.0 / q.M;
            mean = a1 * invM;
            var = a2 * invM - mean * mean;
        } else {
            mean = d2{(double)q.rmean[c], (double)q.rmean[c + 1]};
            var = d2{(double)q.rvar[c], (double)q.rvar[c + 1]};
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ch = c + e;
            const double mu = mean[e], vr = var[e] < 0.0 ? 0.0 : var[e];
            const double rstd = fcn_rsqrt64(vr + (double)q.eps);
            g[e] = q.gamma[ch];
            const double sc = (double)g[e] * rstd;
            s[e] = (float)sc;
            t[e] = (float)((double)q.beta[ch] - mu * sc);
            if (pub) {
                q.bn[ch] = s[e]; q.bn[C3 + ch] = t[e]; q.bn[2 * C3 + ch] = (float)mu; q.bn[3 * C3 + ch] = (float)rstd;
                if (q.stat && q.rmean) {
                    q.rmean[ch] = (float)((1.0 - q.momentum) * q.rmean[ch] + q.momentum * mu);
                    q.rvar[ch] = (float)((1.0 - q.momentum) * q.rvar[ch] + q.momentum * vr * (q.M / (q.M - 1.0)));
                    if (ch == 0 && q.nbt) q.nbt[0] += 1;
                }
            }
        }
    }
    const int l_end = min(L, ((int)blockIdx.x + 1) * WPB);
    for (int l = blockIdx.x * WPB + sub; l < l_end; l += nsub) {
        const int64_t o = ((int64_t)b * L + l) * C3 + c;
        const u32x4 kk = *(const u32x4 *)(pkey + o);               // (two keys as one 16-byte access)
        *(u32x4 *)(pkey + o) = u32x4{0u, 0u, 0u, 0u};
        const unsigned hi[2] = {kk.y, kk.w}, lo[2] = {kk.x, kk.z};
        const bool live = cnt[(int64_t)b * L + l] > 0;           // (an empty window holds one stand-in row: no feature, no gradient)
        float fo[2], yo[2];
        int ao[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int row;
            const float y = fcn_pool_key_value(hi[e], lo[e], g[e], row);
            const float u = fmaf(s[e], y, t[e]);
            const bool pos = live && (hi[e] | lo[e]) != 0u && u > 0.f;
            fo[e] = pos ? u : 0.f;
            ao[e] = pos ? row : -1;
            yo[e] = y;
            // gamma == 0: the key carries no value (fcn_pool_orient stores 0 so that the EARLIEST row wins the tie, as torch.max
            // does on relu(beta)); the backward's sum(dz * xhat) of this channel needs the winner's real y3 -- one gather, only here
            if (ywin && y3 && g[e] == 0.f && pos) {
                const int64_t yi = ((int64_t)b * cap + row) * C3 + c + e;
                yo[e] = s16 ? __uint_as_float((unsigned)((const unsigned short *)y3)[yi] << 16) : ((const float *)y3)[yi];
            }
        }
        *(v2f *)(feat + o) = v2f{fo[0], fo[1]};
        if (amax) { amax[o] = ao[0]; amax[o + 1] = ao[1]; }
        // the winners' pre-BN values, position-major, for the backward's BN3 sums (poolbwd_kernel reads them from the routed-gradient
        // buffer it then overwrites, instead of gathering B*L*C3 single floats out of y3: one dependent round trip and ~36 MB of
        // 64-byte sectors per launch on the widest scale)
        if (ywin) *(v2f *)(ywin + o) = v2f{yo[0], yo[1]};
    }
}

// ------------------------------------------------------------------------------------------------
static inline unsigned pad8(unsigned n) { return (n + 7u) / 8u * 8u; }

template <int MM, int MODE, int POOL>
static int launch_fwd_gemm_mm(const FwdArgs &a, int B, hipStream_t st)
{
    const unsigned nt = (unsigned)(B * a.tps);
    if (FCN_WIDE_TILES && a.COUT % 256 == 0) {
        hipLaunchKernelGGL((fwd_gemm_kernel<MM, MODE, 2, 4, 2, POOL>), dim3(pad8(nt * (a.COUT / 256))), dim3(512), 0, st, a);
    } else if (!FCN_C3_BIG && MODE == 1 && a.COUT >= 512 && a.COUT % 128 == 0) {      // the widest conv3: 64 x 128 tiles
        hipLaunchKernelGGL((fwd_gemm_kernel<MM, MODE, 2, 2, 1, POOL>), dim3(pad8(2 * nt * (a.COUT / 128))), dim3(256), 0, st, a);
    } else if (a.COUT % 128 == 0) {
        hipLaunchKernelGGL((fwd_gemm_kernel<MM, MODE, 2, 2, 2, POOL>), dim3(pad8(nt * (a.COUT / 128))), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((fwd_gemm_kernel<MM, MODE, 1, 2, 2, POOL>), dim3(pad8(nt * (a.COUT / 64))), dim3(256), 0, st, a);
    }
    FCN_CHECK_LAUNCH();
    return 0;
}

template <int MODE>
static int launch_fwd_gemm(const FwdArgs &a, int B, int precision, hipStream_t st)
{
    if (a.CIN % 64 || a.COUT % 64 || a.CIN > MAXC) return FCN_E_BADARG;
    if ((int64_t)B * a.cap * (a.CIN > a.COUT ? a.CIN : a.COUT) >= (int64_t)1 << 31) return FCN_E_LIMIT;      // 32-bit offsets
    if (precision < 0 || precision > FCN_PREC_BF16_OPS) return FCN_E_BADARG;
    if constexpr (MODE == 1) {
        if (a.pkey) {
            if (!a.ewin || !a.gamma_out) return FCN_E_BADARG;
            FCN_MM_SWITCH(FCN_MM_OF(precision, true), return (launch_fwd_gemm_mm<MM, MODE, 1>(a, B, st)));
            return FCN_E_BADARG;
        }
    }
    FCN_MM_SWITCH(FCN_MM_OF(precision, true), return (launch_fwd_gemm_mm<MM, MODE, 0>(a, B, st)));
    return FCN_E_BADARG;
}

extern "C" int fcn_pn_forward(const fcn_pn_desc *d, const fcn_pn_params *p, const int32_t *cnt,
                              const float *one_hot, const fcn_pn_ws *ws, float *feat, void *stream)
{
    if (!d || !p || !ws || !cnt || !feat || !ws->wenc) return FCN_E_BADARG;
    if (d->C1 % 64 || d->C2 % 64 || d->C3 % 64 || d->C1 > MAXC || d->C2 > MAXC) return FCN_E_BADARG;
    if (d->nvec > 0 && !one_hot && !d->nlc) return FCN_E_BADARG;
    if (d->nvec * PW > GT) return FCN_E_LIMIT;
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, L = d->L, K = d->K, C1 = d->C1, C2 = d->C2, C3 = d->C3;
    const int cap = L * K;
    const double M = (double)B * (double)L * (double)K;
    const int tr = d->training ? 1 : 0;
    // a key-pooled training forward hands the winners' pre-BN values to its backward through ws->gmax (fcn_hip.h)
    if (tr && ws->amax && !ws->gmax && fcn_pn_key_pool(d, ws, C3)) return FCN_E_BADARG;
    float *bn1 = ws->bn + fcn_bn_off(0, C1, C2);
    float *bn2 = ws->bn + fcn_bn_off(1, C1, C2);
    float *bn3 = ws->bn + fcn_bn_off(2, C1, C2);
    double *st2 = ws->stat + FCN_STAT_L2, *st3 = st2 + 2 * C2;

    if (!d->grouped) {                  // (fcn_pn_group_compact already finalised BN1 from the input moments and packed the weights)
        FCN_TRY(fcn_pn_pack_weights(d, p, ws, stream));
        hipLaunchKernelGGL(bn1_finalize_kernel, dim3((C1 + 63) / 64), dim3(64), 0, st, ws->stat + FCN_STAT_MOM,
                           p->W[0], p->gamma[0], p->beta[0], p->running_mean[0], p->running_var[0],
                           p->num_batches_tracked[0], C1, tr, d->eps, d->momentum, M, bn1);
        FCN_CHECK_LAUNCH();
    }

    FwdArgs a;
    a.ent = (const float4 *)ws->ent; a.woff = ws->woff; a.tiles = ws->tiles; a.L = L; a.cap = cap; a.tps = (cap + 127) / 128;
    a.aprev = nullptr; a.bn_in = bn1; a.W1 = p->W[0]; a.Wenc = (const u32x4 *)(ws->wenc + pn_wenc_off(0, C1, C2, C3)); a.y = ws->y2;
    a.flags = ws->flags;
    a.stat = tr ? st2 : nullptr; a.CIN = C1; a.COUT = C2;
    a.stat_in = nullptr; a.gamma_in = a.beta_in = nullptr; a.rmean_in = a.rvar_in = nullptr; a.nbt_in = nullptr; a.bn_pub = nullptr;
    a.M = M; a.eps = d->eps; a.momentum = d->momentum; a.rep_stride = 2 * C2 + 2 * C3;
    a.ewin = nullptr; a.pkey = nullptr; a.gamma_out = nullptr;
    FCN_TRY(launch_fwd_gemm<0>(a, B, d->precision, st));

    // BN2 is finalised by conv3's workgroups (no launch in between)
    a.aprev = ws->y2; a.bn_in = bn2; a.W1 = nullptr; a.Wenc = (const u32x4 *)(ws->wenc + pn_wenc_off(1, C1, C2, C3)); a.y = ws->y3;
    a.stat = tr ? st3 : nullptr; a.CIN = C2; a.COUT = C3;
    a.stat_in = tr ? st2 : nullptr; a.gamma_in = p->gamma[1]; a.beta_in = p->beta[1];
    a.rmean_in = p->running_mean[1]; a.rvar_in = p->running_var[1]; a.nbt_in = p->num_batches_tracked[1]; a.bn_pub = bn2;
    // position-major features + a key buffer in the workspace: the max-pool rides in conv3's epilogue (pool_keys_kernel finishes it)
    const bool key_pool = fcn_pn_key_pool(d, ws, C3);
    if (key_pool) { a.ewin = ws->ewin; a.pkey = (unsigned long long *)ws->pkey; a.gamma_out = p->gamma[2]; }
    FCN_TRY(launch_fwd_gemm<1>(a, B, d->precision, st));

    const bool nlc_pool = d->nlc && (C3 == 128 || C3 == 256 || C3 == 512);
    if (!nlc_pool && !key_pool) {        // (the position-major pooling kernels finalise BN3 themselves)
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C3 + 63) / 64), dim3(64), 0, st, st3, 2 * C2 + 2 * C3, p->gamma[2], p->beta[2],
                           p->running_mean[2], p->running_var[2], p->num_batches_tracked[2], C3, tr, d->eps,
                           d->momentum, M, bn3);
        FCN_CHECK_LAUNCH();
    }

    const int nz = FCN_STAT_REP * (2 * C3 + 2 * C2 + 4 * C1);
    const bool s16 = d->precision == FCN_PREC_BF16;       // y2 / y3 stored as bf16 (gemm_tile.h: St)
    if (key_pool) {
        PoolBn pb;
        pb.stat = tr ? st3 : nullptr; pb.rep_stride = 2 * C2 + 2 * C3; pb.gamma = p->gamma[2]; pb.beta = p->beta[2];
        pb.rmean = p->running_mean[2]; pb.rvar = p->running_var[2]; pb.nbt = p->num_batches_tracked[2]; pb.bn = bn3;
        pb.M = M; pb.eps = d->eps; pb.momentum = d->momentum;
        const int nsub = 2 * GT / C3;                       // windows a workgroup covers at once
        int wpb = nsub;
        while ((int64_t)B * ((L + wpb - 1) / wpb) > 1024 && wpb < 8 * nsub) wpb += nsub;       // ~ two workgroups per CU at least
        hipLaunchKernelGGL(pool_keys_kernel, dim3((L + wpb - 1) / wpb, B), dim3(GT), 0, st, (unsigned long long *)ws->pkey, pb, cnt,
                           feat, tr ? ws->amax : nullptr, tr ? ws->gmax : nullptr, L, C3, wpb, tr ? ws->bstat : nullptr, nz,
                           (const void *)ws->y3, cap, s16 ? 1 : 0);
    } else if (nlc_pool) {
        PoolBn pb;
        pb.stat = tr ? st3 : nullptr; pb.rep_stride = 2 * C2 + 2 * C3; pb.gamma = p->gamma[2]; pb.beta = p->beta[2];
        pb.rmean = p->running_mean[2]; pb.rvar = p->running_var[2]; pb.nbt = p->num_batches_tracked[2]; pb.bn = bn3;
        pb.M = M; pb.eps = d->eps; pb.momentum = d->momentum;
        int32_t *am = tr ? ws->amax : nullptr;
        double *zp = tr ? ws->bstat : nullptr;
        // waves per window by the window capacity (nsample): 4 from 128 rows up, else 1 (measured: two waves per window at
        // nsample 64 are SLOWER than one -- 83 -> 96 us for the scale -- the exchange costs more than the shorter walk saves)
#define FCN_POOL_LAUNCH2(VEC_, S16_)                                                                                  \
        if (K >= 128) hipLaunchKernelGGL((pool_nlc_kernel<VEC_, 4, S16_>), dim3(L, 1, B), dim3(GT), 0, st, ws->y3, pb, ws->woff, cnt, feat, am, L, cap, C3, zp, nz); \
        else hipLaunchKernelGGL((pool_nlc_kernel<VEC_, 1, S16_>), dim3((L + PW - 1) / PW, 1, B), dim3(GT), 0, st, ws->y3, pb, ws->woff, cnt, feat, am, L, cap, C3, zp, nz);
#define FCN_POOL_LAUNCH(VEC_) if (s16) { FCN_POOL_LAUNCH2(VEC_, 1) } else { FCN_POOL_LAUNCH2(VEC_, 0) }
        if (C3 == 128) { FCN_POOL_LAUNCH(2) }
        else if (C3 == 256) { FCN_POOL_LAUNCH(4) }
        else { FCN_POOL_LAUNCH(8) }
#undef FCN_POOL_LAUNCH
#undef FCN_POOL_LAUNCH2
    } else {
        dim3 pgrid((L + PW - 1) / PW, C3 / 64, B);
        if (s16) hipLaunchKernelGGL(pool_kernel<1>, pgrid, dim3(GT), 0, st, ws->y3, bn3, ws->woff, cnt, one_hot, feat,
                                    tr ? ws->amax : nullptr, L, cap, C3, d->nvec, d->nlc, tr ? ws->bstat : nullptr, nz);
        else hipLaunchKernelGGL(pool_kernel<0>, pgrid, dim3(GT), 0, st, ws->y3, bn3, ws->woff, cnt, one_hot, feat,
                                tr ? ws->amax : nullptr, L, cap, C3, d->nvec, d->nlc, tr ? ws->bstat : nullptr, nz);
    }
    FCN_CHECK_LAUNCH();
    return 0;
}

// Single-kernel entry (micro-benchmarks, roofline accounting, unit tests): launches only the conv GEMM of
// `layer` (2: conv1+BN1+ReLU fused into conv2; 3: BN2+ReLU fused into conv3) on workspace state left by
// fcn_pn_compact / fcn_pn_forward.  with_stats != 0 also runs the statistics epilogue (accumulates into
// ws->stat, which the caller must not reuse for a backward afterwards).
extern "C" int fcn_pn_conv_fwd(const fcn_pn_desc *d, const fcn_pn_params *p, const fcn_pn_ws *ws, int layer,
                               int with_stats, void *stream)
{
    if (!d || !p || !ws || (layer != 2 && layer != 3) || !ws->wenc) return FCN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;          // (ws->wenc as the forward left it)
    const int B = d->B, L = d->L, K = d->K, C1 = d->C1, C2 = d->C2, C3 = d->C3;
    const int cap = L * K;
    double *st2 = ws->stat + FCN_STAT_L2, *st3 = st2 + 2 * C2;
    FwdArgs a;
    a.ent = (const float4 *)ws->ent; a.woff = ws->woff; a.tiles = ws->tiles; a.L = L; a.cap = cap; a.tps = (cap + 127) / 128;
    a.stat_in = nullptr; a.gamma_in = a.beta_in = nullptr; a.rmean_in = a.rvar_in = nullptr; a.nbt_in = nullptr; a.bn_pub = nullptr;
    a.M = 1.0; a.eps = d->eps; a.momentum = d->momentum;       // the BN in front is read finished from ws->bn
    a.rep_stride = 2 * C2 + 2 * C3;
    a.flags = ws->flags;
    a.ewin = nullptr; a.pkey = nullptr; a.gamma_out = nullptr;
    if (layer == 2) {
        a.aprev = nullptr; a.bn_in = ws->bn + fcn_bn_off(0, C1, C2); a.W1 = p->W[0]; a.Wenc = (const u32x4 *)(ws->wenc + pn_wenc_off(0, C1, C2, C3)); a.y = ws->y2;
        a.stat = with_stats ? st2 : nullptr; a.CIN = C1; a.COUT = C2;
        return launch_fwd_gemm<0>(a, B, d->precision, st);
    }
    a.aprev = ws->y2; a.bn_in = ws->bn + fcn_bn_off(1, C1, C2); a.W1 = nullptr; a.Wenc = (const u32x4 *)(ws->wenc + pn_wenc_off(1, C1, C2, C3)); a.y = ws->y3;
    a.stat = with_stats ? st3 : nullptr; a.CIN = C2; a.COUT = C3;
    return launch_fwd_gemm<1>(a, B, d->precision, st);
}
