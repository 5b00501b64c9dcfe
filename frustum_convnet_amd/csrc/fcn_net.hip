// ConvFeatNet + heads (models/det_base.py:163-224,250-251,365-368) as hand-written HIP: every Conv1d /
// ConvTranspose1d (+ BatchNorm1d + ReLU) is one implicit-GEMM launch on fp32 MFMA over position-major (NLC)
// activations, forward and backward.
//
// The reference runs this part through cuDNN/ATen on (B,C,L) tensors: 13 convs + 13 BN + ReLUs + cats, each its
// own kernel(s); on MI355X the MIOpen path spends most of its time in layout transposes and tiny elementwise
// kernels (profiles/r01_b: 1.9 ms of a 4 ms step).  Here:
//   * activations are rows of C floats (row = (frustum, position)), so a k=3 / stride-2 conv is a GEMM whose A rows
//     are gathered from up to three shifted source rows, a k=1 "merge" conv is a GEMM over concatenated sources
//     (torch.cat never materialises), and ConvTranspose1d with kernel == stride is a plain GEMM with N = k*Cout
//     whose output buffer IS the upsampled NLC tensor;
//   * the producer's BatchNorm + ReLU is applied while the A tile is staged (the conv consumes pre-BN outputs),
//     BN batch statistics are accumulated in the GEMM epilogue;
//   * the one-hot class vector (3 extra input channels broadcast over L) is a virtual 64-channel segment;
//   * backward: dgrad kernels gather dy rows through the inverse position map and write dz of the producer (ReLU
//     mask + BN-backward sums in the epilogue), wgrad kernels reduce over rows in splits + a deterministic reduce
//     that scatters straight into the torch weight layout.
#include "gemm_tile.h"

#define CG_T 256
#define LDC 65                 // k-major LDS leading dim of a 64-wide tile (transposed staging)
#define LDN 68                 // row-major LDS leading dim of a 64-wide tile (float4 aligned)
#define OH_PAD 64              // channels of the virtual one-hot segment
#define CG_SPLIT_ROWS 512      // rows per wgrad split
#define CN_NLAYER 14           // 13 conv/deconv+BN layers + heads

struct CgSeg {
    const float *x;            // type 0: (B*Lsrc, C) rows; type 1: one_hot (B, nvec)
    const float *bn;           // scale[C], shift[C], mean[C], rstd[C] of the producer's BN, or nullptr (already activated)
    int C;                     // channels seen by the GEMM (type 1: OH_PAD)
    int Lsrc;                  // rows per frustum in the source buffer
    int type, nvec;
};

struct CgLayer {
    CgSeg seg[3];
    int nseg;
    int KT, stride, pad;
    int Lin, Lout, B;          // valid input positions, output positions per frustum
    int Cout, Ktot, Cs;        // GEMM N, GEMM K = KT * sum(C), BN channels (col % Cs)
    const float *Wp;           // packed (Cout, Ktot)
    const float *bias;         // (nbias) or nullptr
    int nbias;
    float *y;                  // (B*Lout, Cout) pre-BN output
    double *stat;              // sum[Cs], sumsq[Cs] or nullptr
};

__device__ __forceinline__ void cg_locate(const CgLayer &L, int kk, int &sg, int &tap, int &k0, int &segoff)
{
    sg = 0; segoff = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < L.nseg) {
            const int span = L.KT * L.seg[s].C;
            if (sg == s && kk >= span) { kk -= span; segoff += span; sg = s + 1; }
        }
    }
    tap = kk / L.seg[sg].C;
    k0 = kk % L.seg[sg].C;
}

// A[r][k0..k0+3] of the virtual im2col matrix for output row (b,l) and (segment, tap), activation applied.
__device__ __forceinline__ float4 cg_load_a(const CgLayer &L, int sg, int tap, int kc, int b, int l, bool rvalid)
{
    const CgSeg &S = L.seg[sg];
    const int lin = l * L.stride + tap - L.pad;
    const bool ok = rvalid && lin >= 0 && lin < L.Lin;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok) return v;
    if (S.type == 0) {
        v = *(const float4 *)(S.x + ((int64_t)b * S.Lsrc + lin) * S.C + kc);
        if (S.bn) {
            const float4 s = *(const float4 *)(S.bn + kc), t = *(const float4 *)(S.bn + S.C + kc);
            v.x = fmaxf(fmaf(s.x, v.x, t.x), 0.f); v.y = fmaxf(fmaf(s.y, v.y, t.y), 0.f);
            v.z = fmaxf(fmaf(s.z, v.z, t.z), 0.f); v.w = fmaxf(fmaf(s.w, v.w, t.w), 0.f);
        }
    } else {
        const float *oh = S.x + (int64_t)b * S.nvec;
        v.x = (kc + 0 < S.nvec) ? oh[kc + 0] : 0.f; v.y = (kc + 1 < S.nvec) ? oh[kc + 1] : 0.f;
        v.z = (kc + 2 < S.nvec) ? oh[kc + 2] : 0.f; v.w = (kc + 3 < S.nvec) ? oh[kc + 3] : 0.f;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// Forward: y[r][n] = sum_kk A[r][kk] * Wp[n][kk] (+bias); 64 x 64 tile, 4 waves of one 32x32 MFMA tile each.
// gridDim.z > 1: split-K -- split z reduces chunks [z*cps, (z+1)*cps) and stores its raw tile into `partial`
// (S, R, Cout); cg_fwd_finish_kernel sums the splits and runs the epilogue.  These GEMMs are small (B*L rows): without
// the split a layer is ~140 workgroups of up to 48 dependent chunk iterations on 256 CUs (latency-bound, 37 us/layer).
__global__ __launch_bounds__(CG_T) void cg_fwd_kernel(CgLayer L, float *__restrict__ partial, int cps)
{
    __shared__ float As[KC * LDC];
    __shared__ float Bs[KC * LDC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int R = L.B * L.Lout;
    const int row0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int kq = tid & 7, rb = tid >> 3;
    int bb[2], ll[2];
    bool rv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = row0 + rb + 32 * i;
        rv[i] = g < R;
        bb[i] = rv[i] ? g / L.Lout : 0;
        ll[i] = rv[i] ? g % L.Lout : 0;
    }
    f32x16 acc[1][1];
    acc_zero<1, 1>(acc);
    float4 ra[2], rw[2];
    const int nchunk_all = L.Ktot / KC;
    const int cbeg = blockIdx.z * cps, nchunk = min(nchunk_all, cbeg + cps);
    auto load_chunk = [&](int c) {
        int sg, tap, k0, so;
        cg_locate(L, c * KC, sg, tap, k0, so);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = cg_load_a(L, sg, tap, k0 + 4 * kq, bb[i], ll[i], rv[i]);
            rw[i] = *(const float4 *)(L.Wp + (int64_t)(n0 + rb + 32 * i) * L.Ktot + c * KC + 4 * kq);
        }
    };
    load_chunk(cbeg);
    for (int c = cbeg; c < nchunk; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = rb + 32 * i;
            As[(4 * kq + 0) * LDC + r] = ra[i].x; As[(4 * kq + 1) * LDC + r] = ra[i].y;
            As[(4 * kq + 2) * LDC + r] = ra[i].z; As[(4 * kq + 3) * LDC + r] = ra[i].w;
            Bs[(4 * kq + 0) * LDC + r] = rw[i].x; Bs[(4 * kq + 1) * LDC + r] = rw[i].y;
            Bs[(4 * kq + 2) * LDC + r] = rw[i].z; Bs[(4 * kq + 3) * LDC + r] = rw[i].w;
        }
        __syncthreads();
        if (c + 1 < nchunk) load_chunk(c + 1);
        mma_chunk<1, 1, LDC, LDC>(As, Bs, wm * 32, wn * 32, acc);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l31;
    if (partial) {
        float *pp = partial + (int64_t)blockIdx.z * R * L.Cout;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = row0 + wm * 32 + acc_row(reg, lh);
            if (row < R) pp[(int64_t)row * L.Cout + col] = acc[0][0][reg];
        }
        return;
    }
    const float bias = (L.bias && col < L.nbias) ? L.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = row0 + wm * 32 + acc_row(reg, lh);
        if (row < R) {
            const float v = acc[0][0][reg] + bias;
            L.y[(int64_t)row * L.Cout + col] = v;
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    }
    if (L.stat) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float *red = As;
        if (wm == 1 && lh == 0) { red[(wn * 32 + l31) * 2] = s1; red[(wn * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (wm == 0 && lh == 0) {
            const int ch = col % L.Cs;
            atomic_add_f64(&L.stat[ch], (double)s1 + (double)red[(wn * 32 + l31) * 2]);
            atomic_add_f64(&L.stat[L.Cs + ch], (double)s2 + (double)red[(wn * 32 + l31) * 2 + 1]);
        }
    }
}

// y = sum of the split partials (+bias), BN statistics.  Workgroup: 128 rows x 64 columns, thread = 4 columns x 16 row lanes.
__global__ __launch_bounds__(CG_T) void cg_fwd_finish_kernel(CgLayer L, const float *__restrict__ partial, int S)
{
    __shared__ float red[16][64][2];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int R = L.B * L.Lout;
    const int col = blockIdx.y * 64 + 4 * cq;
    const int rbeg = blockIdx.x * 128, rend = min(R, rbeg + 128);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.bias) {
        bias.x = col + 0 < L.nbias ? L.bias[col + 0] : 0.f; bias.y = col + 1 < L.nbias ? L.bias[col + 1] : 0.f;
        bias.z = col + 2 < L.nbias ? L.bias[col + 2] : 0.f; bias.w = col + 3 < L.nbias ? L.bias[col + 3] : 0.f;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t plane = (int64_t)R * L.Cout;
    for (int row = rbeg + rl; row < rend; row += 16) {
        const int64_t o = (int64_t)row * L.Cout + col;
        float4 v = bias;
        for (int sp = 0; sp < S; ++sp) {
            const float4 q = *(const float4 *)(partial + sp * plane + o);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        *(float4 *)(L.y + o) = v;
        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
        s2[0] = fmaf(v.x, v.x, s2[0]); s2[1] = fmaf(v.y, v.y, s2[1]);
        s2[2] = fmaf(v.z, v.z, s2[2]); s2[3] = fmaf(v.w, v.w, s2[3]);
    }
    if (!L.stat) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][4 * cq + j][0] = s1[j]; red[rl][4 * cq + j][1] = s2[j]; }
    __syncthreads();
    if (tid < 128) {
        const int c = tid & 63, w = tid >> 6;
        double a = 0.0;
        for (int r = 0; r < 16; ++r) a += (double)red[r][c][w];
        const int ch = (blockIdx.y * 64 + c) % L.Cs;
        atomic_add_f64(&L.stat[w * L.Cs + ch], a);
    }
}

// BN scale/shift/mean/rstd (+ running stats) from sum / sumsq over M rows; eval mode reads the running stats.
__global__ void cn_bn_finalize_kernel(const double *__restrict__ stat, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float *rmean, float *rvar, int64_t *nbt,
                                      int C, int training, float eps, float momentum, double M, float *__restrict__ bn)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = stat[c] / M;
        var = stat[C + c] / M - mean * mean;
        if (var < 0.0) var = 0.0;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * (M / (M - 1.0)));
        if (c == 0) nbt[0] += 1;
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

// coef: gamma*rstd, mean, rstd, dbeta/M, dgamma/M ; exports dgamma, dbeta
__global__ void cn_bnbwd_finalize_kernel(const double *__restrict__ bstat, const float *__restrict__ gamma,
                                         const float *__restrict__ bn, int C, double M, float *__restrict__ coef,
                                         float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double db = bstat[c], dg = bstat[C + c];
    const float rstd = bn[3 * C + c];
    coef[c] = gamma[c] * rstd;
    coef[C + c] = bn[2 * C + c];
    coef[2 * C + c] = rstd;
    coef[3 * C + c] = (float)(db / M);
    coef[4 * C + c] = (float)(dg / M);
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
}

// dy of a BN layer for 4 consecutive columns: kk*(dz - dbeta/M - xhat*dgamma/M); coef == nullptr: dy = dz (heads)
__device__ __forceinline__ float4 cg_dy4(const float *dz, const float *y, const float *coef, int Cs, int64_t off, int col)
{
    float4 d = *(const float4 *)(dz + off);
    if (!coef) return d;
    const float4 yv = *(const float4 *)(y + off);
    const int ch = col % Cs;                 // 4 consecutive columns never straddle a Cs boundary (Cs % 64 == 0)
    const float4 kk = *(const float4 *)(coef + ch), mu = *(const float4 *)(coef + Cs + ch);
    const float4 rs = *(const float4 *)(coef + 2 * Cs + ch), cb = *(const float4 *)(coef + 3 * Cs + ch);
    const float4 cg = *(const float4 *)(coef + 4 * Cs + ch);
    d.x = kk.x * (d.x - fmaf((yv.x - mu.x) * rs.x, cg.x, cb.x));
    d.y = kk.y * (d.y - fmaf((yv.y - mu.y) * rs.y, cg.y, cb.y));
    d.z = kk.z * (d.z - fmaf((yv.z - mu.z) * rs.z, cg.z, cb.z));
    d.w = kk.w * (d.w - fmaf((yv.w - mu.w) * rs.w, cg.w, cb.w));
    return d;
}

// ------------------------------------------------------------------------------------------------
struct CgDgrad {
    CgLayer lay;               // the consumer layer
    int sg, segoff;            // which of its segments is differentiated; its column offset in Wp
    const float *dzc, *yc, *coefc;   // consumer's incoming dz, pre-BN output, BN-backward coefficients (null: no BN)
    const float *ysrc, *bnsrc; // producer's pre-BN output and BN (scale,shift,mean,rstd); null: source is a plain input
    float *out;                // producer's dz (Rsrc x C) or the input gradient
    int accumulate;            // add to what `out` already holds (a second consumer)
    double *bstat_src;         // non-null on the LAST consumer: sum dz, sum dz*xhat of the producer
};

// G[rs][k] = sum_tap sum_n dy[r_out(rs,tap)][n] * Wp[n][segoff + tap*C + k]; 64 source rows x 64 source channels.
__global__ __launch_bounds__(CG_T) void cg_dgrad_kernel(CgDgrad a, float *__restrict__ partial, int cps)
{
    __shared__ __attribute__((aligned(16))) float As[KC * LDC];
    __shared__ __attribute__((aligned(16))) float Bs[KC * LDN];
    const CgLayer &L = a.lay;
    const CgSeg &S = L.seg[a.sg];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int C = S.C, Rs = L.B * S.Lsrc;
    const int row0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int kq = tid & 7, rb = tid >> 3;
    int bb[2], li[2];
    bool rv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = row0 + rb + 32 * i;
        rv[i] = g < Rs;
        bb[i] = rv[i] ? g / S.Lsrc : 0;
        li[i] = rv[i] ? g % S.Lsrc : 0;
        rv[i] = rv[i] && li[i] < L.Lin;          // cropped positions receive no gradient
    }
    f32x16 acc[1][1];
    acc_zero<1, 1>(acc);
    float4 ra[2], rw[2];
    const int ncn = L.Cout / KC;                  // n-chunks per tap
    const int cbeg = blockIdx.z * cps, nchunk = min(L.KT * ncn, cbeg + cps);
    auto load_chunk = [&](int c) {
        const int tap = c / ncn, nq = (c % ncn) * KC + 4 * kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = li[i] + L.pad - tap;
            const int lo = t / L.stride;
            const bool ok = rv[i] && t >= 0 && (t % L.stride) == 0 && lo < L.Lout;
            ra[i] = ok ? cg_dy4(a.dzc, a.yc, a.coefc, L.Cs, ((int64_t)bb[i] * L.Lout + lo) * L.Cout + nq, nq)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + CG_T * i;
            const int nn = f >> 4, cq = f & 15;
            rw[i] = *(const float4 *)(L.Wp + (int64_t)((c % ncn) * KC + nn) * L.Ktot + a.segoff + tap * C + c0 + 4 * cq);
        }
    };
    load_chunk(cbeg);
    for (int c = cbeg; c < nchunk; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = rb + 32 * i;
            As[(4 * kq + 0) * LDC + r] = ra[i].x; As[(4 * kq + 1) * LDC + r] = ra[i].y;
            As[(4 * kq + 2) * LDC + r] = ra[i].z; As[(4 * kq + 3) * LDC + r] = ra[i].w;
            const int f = tid + CG_T * i;
            *(float4 *)(Bs + (f >> 4) * LDN + 4 * (f & 15)) = rw[i];
        }
        __syncthreads();
        if (c + 1 < nchunk) load_chunk(c + 1);
        mma_chunk<1, 1, LDC, LDN>(As, Bs, wm * 32, wn * 32, acc);
        __syncthreads();
    }
    const int col = c0 + wn * 32 + l31;
    if (partial) {
        float *pp = partial + (int64_t)blockIdx.z * Rs * C;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = row0 + wm * 32 + acc_row(reg, lh);
            if (row < Rs) pp[(int64_t)row * C + col] = acc[0][0][reg];
        }
        return;
    }
    float ps = 0.f, pt = 0.f, pm = 0.f, pr = 0.f;
    if (a.bnsrc) { ps = a.bnsrc[col]; pt = a.bnsrc[C + col]; pm = a.bnsrc[2 * C + col]; pr = a.bnsrc[3 * C + col]; }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = row0 + wm * 32 + acc_row(reg, lh);
        if (row < Rs) {
            const int64_t o = (int64_t)row * C + col;
            float g = acc[0][0][reg];
            if (a.bnsrc) {
                const float yv = a.ysrc[o];
                g = (fmaf(ps, yv, pt) > 0.f) ? g : 0.f;
                if (a.accumulate) g += a.out[o];
                a.out[o] = g;
                s1 += g;
                s2 = fmaf(g, (yv - pm) * pr, s2);
            } else {
                if (a.accumulate) g += a.out[o];
                a.out[o] = g;
            }
        }
    }
    if (a.bstat_src) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float *red = As;
        if (wm == 1 && lh == 0) { red[(wn * 32 + l31) * 2] = s1; red[(wn * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (wm == 0 && lh == 0) {
            atomic_add_f64(&a.bstat_src[col], (double)s1 + (double)red[(wn * 32 + l31) * 2]);
            atomic_add_f64(&a.bstat_src[C + col], (double)s2 + (double)red[(wn * 32 + l31) * 2 + 1]);
        }
    }
}

// Sum of the dgrad split partials + the producer-side epilogue (ReLU mask, accumulate, BN-backward sums).
__global__ __launch_bounds__(CG_T) void cg_dgrad_finish_kernel(CgDgrad a, const float *__restrict__ partial, int S)
{
    __shared__ float red[16][64][2];
    const CgSeg &Sg = a.lay.seg[a.sg];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int C = Sg.C, Rs = a.lay.B * Sg.Lsrc;
    const int col = blockIdx.y * 64 + 4 * cq;
    const int rbeg = blockIdx.x * 128, rend = min(Rs, rbeg + 128);
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pt = ps, pm = ps, pr = ps;
    if (a.bnsrc) {
        ps = *(const float4 *)(a.bnsrc + col); pt = *(const float4 *)(a.bnsrc + C + col);
        pm = *(const float4 *)(a.bnsrc + 2 * C + col); pr = *(const float4 *)(a.bnsrc + 3 * C + col);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t plane = (int64_t)Rs * C;
    for (int row = rbeg + rl; row < rend; row += 16) {
        const int64_t o = (int64_t)row * C + col;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < S; ++sp) {
            const float4 q = *(const float4 *)(partial + sp * plane + o);
            g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w;
        }
        if (a.bnsrc) {
            const float4 yv = *(const float4 *)(a.ysrc + o);
            g.x = fmaf(ps.x, yv.x, pt.x) > 0.f ? g.x : 0.f; g.y = fmaf(ps.y, yv.y, pt.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(ps.z, yv.z, pt.z) > 0.f ? g.z : 0.f; g.w = fmaf(ps.w, yv.w, pt.w) > 0.f ? g.w : 0.f;
            if (a.accumulate) {
                const float4 p0 = *(const float4 *)(a.out + o);
                g.x += p0.x; g.y += p0.y; g.z += p0.z; g.w += p0.w;
            }
            *(float4 *)(a.out + o) = g;
            s1[0] += g.x; s1[1] += g.y; s1[2] += g.z; s1[3] += g.w;
            s2[0] = fmaf(g.x, (yv.x - pm.x) * pr.x, s2[0]); s2[1] = fmaf(g.y, (yv.y - pm.y) * pr.y, s2[1]);
            s2[2] = fmaf(g.z, (yv.z - pm.z) * pr.z, s2[2]); s2[3] = fmaf(g.w, (yv.w - pm.w) * pr.w, s2[3]);
        } else {
            if (a.accumulate) {
                const float4 p0 = *(const float4 *)(a.out + o);
                g.x += p0.x; g.y += p0.y; g.z += p0.z; g.w += p0.w;
            }
            *(float4 *)(a.out + o) = g;
        }
    }
    if (!a.bstat_src) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][4 * cq + j][0] = s1[j]; red[rl][4 * cq + j][1] = s2[j]; }
    __syncthreads();
    if (tid < 128) {
        const int c = tid & 63, w = tid >> 6;
        double v = 0.0;
        for (int r = 0; r < 16; ++r) v += (double)red[r][c][w];
        atomic_add_f64(&a.bstat_src[w * C + blockIdx.y * 64 + c], v);
    }
}

// ------------------------------------------------------------------------------------------------
struct CgWgrad {
    CgLayer lay;
    const float *dz, *coef;    // incoming dz of this layer (R x Cout), BN-backward coefficients (null: no BN)
    float *partial;            // (nsplit, Cout, Ktot)
    int rows;                  // rows per split (multiple of 32)
};

// dWp[n][kk] = sum_r dy[r][n] * A[r][kk]; tile 64 (n) x 64 (kk), rows split in CG_SPLIT_ROWS.
__global__ __launch_bounds__(CG_T) void cg_wgrad_kernel(CgWgrad a)
{
    __shared__ __attribute__((aligned(16))) float As[KC * LDN];
    __shared__ __attribute__((aligned(16))) float Bs[KC * LDN];
    const CgLayer &L = a.lay;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int R = L.B * L.Lout;
    const int rbeg = blockIdx.x * a.rows, rend = min(R, rbeg + a.rows);
    const int n0 = blockIdx.y * 64, kk0 = blockIdx.z * 64;
    int sg, tap, k0, so;
    cg_locate(L, kk0, sg, tap, k0, so);
    const int cq = tid & 15, rr0 = tid >> 4;              // column quad, first row (rows rr0, rr0+16)
    f32x16 acc[1][1];
    acc_zero<1, 1>(acc);
    float4 ra[2], rb4[2];
    auto load_chunk = [&](int r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + rr0 + 16 * i;
            if (row < rend) {
                ra[i] = cg_dy4(a.dz, L.y, a.coef, L.Cs, (int64_t)row * L.Cout + n0 + 4 * cq, n0 + 4 * cq);
                rb4[i] = cg_load_a(L, sg, tap, k0 + 4 * cq, row / L.Lout, row % L.Lout, true);
            } else {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                rb4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    load_chunk(rbeg);
    for (int r0 = rbeg; r0 < rend; r0 += KC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(float4 *)(As + (rr0 + 16 * i) * LDN + 4 * cq) = ra[i];
            *(float4 *)(Bs + (rr0 + 16 * i) * LDN + 4 * cq) = rb4[i];
        }
        __syncthreads();
        if (r0 + KC < rend) load_chunk(r0 + KC);
        mma_chunk<1, 1, LDN, LDN>(As, Bs, wm * 32, wn * 32, acc);
        __syncthreads();
    }
    float *out = a.partial + (int64_t)blockIdx.x * L.Cout * L.Ktot;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wm * 32 + acc_row(reg, lh);
        const int kk = kk0 + wn * 32 + l31;
        out[(int64_t)n * L.Ktot + kk] = acc[0][0][reg];
    }
}

// Weight packing descriptor of one layer: conv (Cout, Cin, KT) or deconv (Cin, Cout, k) <-> packed (N, Ktot).
struct CgPack {
    int N, Ktot, KT, nseg;
    int C[3], choff[3], type[3];      // GEMM channels, channel offset in the torch weight, segment type
    int nvec, cin_tot, deconv_k, cout_t;
};

__device__ __forceinline__ int64_t cg_torch_index(const CgPack &p, int n, int kk)
{
    if (p.deconv_k > 0) {               // row n = j*Cout + co, kk = ci  ->  W[ci][co][j]
        const int j = n / p.cout_t, co = n % p.cout_t;
        return ((int64_t)kk * p.cout_t + co) * p.deconv_k + j;
    }
    int sg = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s)
        if (s < p.nseg && sg == s && kk >= p.KT * p.C[s]) { kk -= p.KT * p.C[s]; sg = s + 1; }
    const int tap = kk / p.C[sg], k = kk % p.C[sg];
    if (p.type[sg] == 1 && k >= p.nvec) return -1;       // padding column of the virtual one-hot segment
    return ((int64_t)n * p.cin_tot + p.choff[sg] + k) * p.KT + tap;
}

struct CgPackAll {
    CgPack p[CN_NLAYER];
    const float *src[CN_NLAYER];
    float *dst[CN_NLAYER];
    int64_t pre[CN_NLAYER + 1];
    int nrow_real[CN_NLAYER];           // rows of the packed matrix that exist in the torch weight (heads: 41 of 64)
};

__global__ void cg_pack_kernel(CgPackAll t)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.pre[CN_NLAYER]) return;
    int l = 0;
#pragma unroll
    for (int q = 1; q < CN_NLAYER; ++q)
        if (i >= t.pre[q]) l = q;
    const int64_t e = i - t.pre[l];
    const int n = (int)(e / t.p[l].Ktot), kk = (int)(e % t.p[l].Ktot);
    float v = 0.f;
    if (n < t.nrow_real[l]) {
        const int64_t o = cg_torch_index(t.p[l], n, kk);
        if (o >= 0) v = t.src[l][o];
    }
    t.dst[l][e] = v;
}

// dW (torch layout) = sum of the split partials; padding columns/rows are dropped.
__global__ void cg_wgrad_reduce_kernel(const float *__restrict__ partial, int nsplit, CgPack p, int nrow_real,
                                       float *__restrict__ dW)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nelem = (int64_t)p.N * p.Ktot;
    if (e >= nelem) return;
    const int n = (int)(e / p.Ktot), kk = (int)(e % p.Ktot);
    if (n >= nrow_real) return;
    const int64_t o = cg_torch_index(p, n, kk);
    if (o < 0) return;
    float s0 = 0.f, s1 = 0.f;
    int sp = 0;
    for (; sp + 2 <= nsplit; sp += 2) { s0 += partial[(int64_t)sp * nelem + e]; s1 += partial[(int64_t)(sp + 1) * nelem + e]; }
    if (sp < nsplit) s0 += partial[(int64_t)sp * nelem + e];
    dW[o] = s0 + s1;
}

// dbias[n] = sum_r dlogits[r][n] (heads)
__global__ void cg_colsum_kernel(const float *__restrict__ d, int R, int ld, int ncol, float *__restrict__ out)
{
    __shared__ float sh[256];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) s += d[(int64_t)r * ld + n];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && n < ncol) out[n] = sh[0];
}

// ================================================================================================
// Host side: the fixed topology of ConvFeatNet(128, nvec) + heads.
// layer ids:  0 b1c1  1 b2c1  2 b2c2  3 b2m  4 b3c1  5 b3c2  6 b3m  7 b4c1  8 b4c2  9 b4m  10 b2d  11 b3d  12 b4d  13 heads
// ================================================================================================
struct CnPlan {
    int KT[CN_NLAYER], stride[CN_NLAYER], pad[CN_NLAYER], Lin[CN_NLAYER], Lout[CN_NLAYER];
    int N[CN_NLAYER], Cs[CN_NLAYER], Ktot[CN_NLAYER], dk[CN_NLAYER];
    int nseg[CN_NLAYER], src[CN_NLAYER][3], C[CN_NLAYER][3], choff[CN_NLAYER][3];   // src: layer id, -1..-4 feats, -9 one-hot
    int cin_tot[CN_NLAYER], nrow_real[CN_NLAYER];
};

static int conv_len(int L, int k, int s, int p) { return (L + 2 * p - k) / s + 1; }

// chunks per split so that tiles*splits ~ 1024 workgroups and every split keeps >= 2 chunks
static int pick_cps(int tiles, int nchunk)
{
    int S = 1024 / (tiles > 0 ? tiles : 1);
    if (S > nchunk / 2) S = nchunk / 2;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    return (nchunk + S - 1) / S;
}

// rows per wgrad split (multiple of 32): ~1024 workgroups, >= 64 rows each
static int pick_wrows(int R, int out_tiles)
{
    int ns = 1024 / (out_tiles > 0 ? out_tiles : 1);
    if (ns > (R + 63) / 64) ns = (R + 63) / 64;
    if (ns < 1) ns = 1;
    int rows = (R + ns - 1) / ns;
    rows = (rows + 31) / 32 * 32;
    return rows;
}

static int cn_make_plan(const fcn_cn_desc *d, CnPlan &P)
{
    const int L1 = d->L[0], L2 = d->L[1], L3 = d->L[2], L4 = d->L[3];
    if (conv_len(L1, 3, 2, 1) != L2 || conv_len(L2, 3, 2, 1) != L3 || conv_len(L3, 3, 2, 1) != L4) return FCN_E_BADARG;
    if (2 * L3 < L2 || 4 * L4 < L2) return FCN_E_BADARG;
    const int cw[10] = {128, 128, 128, 128, 256, 256, 256, 512, 512, 512};
    const int Lo[10] = {L1, L2, L2, L2, L3, L3, L3, L4, L4, L4};
    const int Li[10] = {L1, L1, L2, L2, L2, L3, L3, L3, L4, L4};
    const int kt[10] = {3, 3, 3, 1, 3, 3, 1, 3, 3, 1};
    const int st[10] = {1, 2, 1, 1, 2, 1, 1, 2, 1, 1};
    for (int l = 0; l < CN_NLAYER; ++l) { P.dk[l] = 0; P.nseg[l] = 1; for (int s = 0; s < 3; ++s) { P.src[l][s] = 0; P.C[l][s] = 0; P.choff[l][s] = 0; } }
    for (int l = 0; l < 10; ++l) {
        P.KT[l] = kt[l]; P.stride[l] = st[l]; P.pad[l] = kt[l] == 3 ? 1 : 0; P.Lin[l] = Li[l]; P.Lout[l] = Lo[l];
        P.N[l] = cw[l]; P.Cs[l] = cw[l]; P.nrow_real[l] = cw[l];
    }
    // segments
    P.nseg[0] = 2; P.src[0][0] = -1; P.C[0][0] = 128; P.src[0][1] = -9; P.C[0][1] = OH_PAD; P.choff[0][1] = 128;
    P.src[1][0] = 0; P.C[1][0] = 128;  P.src[2][0] = 1; P.C[2][0] = 128;
    P.nseg[3] = 3; P.src[3][0] = 2; P.C[3][0] = 128; P.src[3][1] = -2; P.C[3][1] = 128; P.choff[3][1] = 128;
    P.src[3][2] = -9; P.C[3][2] = OH_PAD; P.choff[3][2] = 256;
    P.src[4][0] = 3; P.C[4][0] = 128;  P.src[5][0] = 4; P.C[5][0] = 256;
    P.nseg[6] = 3; P.src[6][0] = 5; P.C[6][0] = 256; P.src[6][1] = -3; P.C[6][1] = 256; P.choff[6][1] = 256;
    P.src[6][2] = -9; P.C[6][2] = OH_PAD; P.choff[6][2] = 512;
    P.src[7][0] = 6; P.C[7][0] = 256;  P.src[8][0] = 7; P.C[8][0] = 512;
    P.nseg[9] = 3; P.src[9][0] = 8; P.C[9][0] = 512; P.src[9][1] = -4; P.C[9][1] = 512; P.choff[9][1] = 512;
    P.src[9][2] = -9; P.C[9][2] = OH_PAD; P.choff[9][2] = 1024;
    // deconvs: GEMM over input rows with N = k * 256
    const int dsrc[3] = {3, 6, 9}, dkk[3] = {1, 2, 4}, dci[3] = {128, 256, 512}, dL[3] = {L2, L3, L4};
    for (int q = 0; q < 3; ++q) {
        const int l = 10 + q;
        P.KT[l] = 1; P.stride[l] = 1; P.pad[l] = 0; P.Lin[l] = dL[q]; P.Lout[l] = dL[q];
        P.N[l] = dkk[q] * 256; P.Cs[l] = 256; P.dk[l] = dkk[q]; P.nrow_real[l] = P.N[l];
        P.src[l][0] = dsrc[q]; P.C[l][0] = dci[q];
    }
    // heads over cat(xx1, xx2[:L2], xx3[:L2]) -> 2 + out_size columns, padded to 64
    P.KT[13] = 1; P.stride[13] = 1; P.pad[13] = 0; P.Lin[13] = L2; P.Lout[13] = L2; P.N[13] = 64; P.Cs[13] = 64;
    P.nrow_real[13] = 2 + d->reg_out; P.nseg[13] = 3;
    for (int s = 0; s < 3; ++s) { P.src[13][s] = 10 + s; P.C[13][s] = 256; P.choff[13][s] = 256 * s; }
    if (P.nrow_real[13] > 64) return FCN_E_LIMIT;
    for (int l = 0; l < CN_NLAYER; ++l) {
        int cs = 0, ct = 0;
        for (int s = 0; s < P.nseg[l]; ++s) { cs += P.C[l][s]; ct += (P.src[l][s] == -9) ? d->nvec : P.C[l][s]; }
        P.Ktot[l] = P.KT[l] * cs;
        P.cin_tot[l] = ct;
    }
    return 0;
}

// element offsets of each layer inside the shared workspaces
struct CnOffsets {
    int64_t y[CN_NLAYER + 1], wp[CN_NLAYER + 1];
    int bn[CN_NLAYER + 1], st[CN_NLAYER + 1], coef[CN_NLAYER + 1];
};

static void cn_offsets(const fcn_cn_desc *d, const CnPlan &P, CnOffsets &O)
{
    O.y[0] = O.wp[0] = 0; O.bn[0] = O.st[0] = O.coef[0] = 0;
    for (int l = 0; l < CN_NLAYER; ++l) {
        O.y[l + 1] = O.y[l] + (int64_t)d->B * P.Lout[l] * P.N[l];
        O.wp[l + 1] = O.wp[l] + (int64_t)P.N[l] * P.Ktot[l];
        O.bn[l + 1] = O.bn[l] + 4 * P.Cs[l];
        O.st[l + 1] = O.st[l] + 2 * P.Cs[l];
        O.coef[l + 1] = O.coef[l] + 5 * P.Cs[l];
    }
}

extern "C" int fcn_convnet_sizes(const fcn_cn_desc *d, int64_t *out6)
{
    if (!d || !out6) return FCN_E_BADARG;
    CnPlan P;
    FCN_TRY(cn_make_plan(d, P));
    CnOffsets O;
    cn_offsets(d, P, O);
    int64_t pmax = 0;
    for (int l = 0; l < CN_NLAYER; ++l) {
        const int64_t R = (int64_t)d->B * P.Lout[l];
        {   // forward split-K partials
            const int tiles = (int)((R + 63) / 64) * (P.N[l] / 64), nch = P.Ktot[l] / KC;
            const int cps = pick_cps(tiles, nch), S = (nch + cps - 1) / cps;
            const int64_t v = (int64_t)S * R * P.N[l];
            if (v > pmax) pmax = v;
        }
        {   // wgrad row splits
            const int rows = pick_wrows((int)R, (P.N[l] / 64) * (P.Ktot[l] / 64));
            const int64_t v = ((R + rows - 1) / rows) * P.N[l] * P.Ktot[l];
            if (v > pmax) pmax = v;
        }
        for (int sg = 0; sg < P.nseg[l]; ++sg) {   // dgrad split partials
            if (P.src[l][sg] == -9) continue;
            const int src = P.src[l][sg];
            const int Lsrc = src < 0 ? d->L[-src - 1] : P.Lout[src] * (P.dk[src] > 0 ? P.dk[src] : 1);
            const int64_t Rs = (int64_t)d->B * Lsrc;
            const int tiles = (int)((Rs + 63) / 64) * (P.C[l][sg] / 64), nch = P.KT[l] * P.N[l] / KC;
            const int cps = pick_cps(tiles, nch), S = (nch + cps - 1) / cps;
            const int64_t v = (int64_t)S * Rs * P.C[l][sg];
            if (v > pmax) pmax = v;
        }
    }
    out6[0] = O.y[CN_NLAYER];      // floats: y (and dz) of all layers
    out6[1] = O.wp[CN_NLAYER];     // floats: packed weights
    out6[2] = O.bn[CN_NLAYER];     // floats: bn scale/shift/mean/rstd
    out6[3] = O.st[CN_NLAYER];     // doubles: stat (and bstat)
    out6[4] = O.coef[CN_NLAYER];   // floats: coef
    out6[5] = pmax;                // floats: wgrad partials
    return 0;
}

static void cn_fill_layer(const fcn_cn_desc *d, const CnPlan &P, const CnOffsets &O, const fcn_cn_ws *ws,
                          const float *const feats[4], const float *one_hot, int l, CgLayer &L)
{
    L.nseg = P.nseg[l]; L.KT = P.KT[l]; L.stride = P.stride[l]; L.pad = P.pad[l];
    L.Lin = P.Lin[l]; L.Lout = P.Lout[l]; L.B = d->B; L.Cout = P.N[l]; L.Ktot = P.Ktot[l]; L.Cs = P.Cs[l];
    L.Wp = ws->wp + O.wp[l]; L.bias = nullptr; L.nbias = 0; L.y = ws->y + O.y[l]; L.stat = nullptr;
    for (int s = 0; s < 3; ++s) {
        CgSeg &S = L.seg[s];
        S.x = nullptr; S.bn = nullptr; S.C = P.C[l][s]; S.Lsrc = P.Lin[l]; S.type = 0; S.nvec = 0;
        if (s >= P.nseg[l]) continue;
        const int src = P.src[l][s];
        if (src == -9) { S.type = 1; S.x = one_hot; S.nvec = d->nvec; }
        else if (src < 0) { S.x = feats[-src - 1]; S.Lsrc = d->L[-src - 1]; }
        else {
            S.x = ws->y + O.y[src]; S.bn = ws->bn + O.bn[src];
            S.Lsrc = P.Lout[src] * (P.dk[src] > 0 ? P.dk[src] : 1);     // a deconv's buffer is (B, L*k, 256)
        }
    }
}

static void cn_fill_pack(const fcn_cn_desc *d, const CnPlan &P, int l, CgPack &p)
{
    p.N = P.N[l]; p.Ktot = P.Ktot[l]; p.KT = P.KT[l]; p.nseg = P.nseg[l];
    for (int s = 0; s < 3; ++s) { p.C[s] = P.C[l][s]; p.choff[s] = P.choff[l][s]; p.type[s] = (P.src[l][s] == -9) ? 1 : 0; }
    p.nvec = d->nvec; p.cin_tot = P.cin_tot[l]; p.deconv_k = P.dk[l]; p.cout_t = 256;
}

extern "C" int fcn_convnet_forward(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws,
                                   const float *const feats[4], const float *one_hot, float *logits, void *stream)
{
    if (!d || !p || !ws || !feats || !logits) return FCN_E_BADARG;
    if (!ws->y || !ws->wp || !ws->bn || !ws->stat || !ws->partial) return FCN_E_BADARG;
    if (d->nvec > 0 && !one_hot) return FCN_E_BADARG;
    if (d->nvec > OH_PAD) return FCN_E_LIMIT;
    hipStream_t st = (hipStream_t)stream;
    CnPlan P;
    FCN_TRY(cn_make_plan(d, P));
    CnOffsets O;
    cn_offsets(d, P, O);
    const int tr = d->training ? 1 : 0;
    if (tr) {
        hipError_t e = hipMemsetAsync(ws->stat, 0, sizeof(double) * (size_t)O.st[CN_NLAYER], st);
        if (e != hipSuccess) return (int)e;
    }
    // pack all weights in one launch (heads: rows 0..1 cls_out, 2.. reg_out)
    CgPackAll t;
    t.pre[0] = 0;
    for (int l = 0; l < CN_NLAYER; ++l) {
        cn_fill_pack(d, P, l, t.p[l]);
        t.src[l] = p->W[l]; t.dst[l] = ws->wp + O.wp[l];
        t.pre[l + 1] = t.pre[l] + (int64_t)P.N[l] * P.Ktot[l];
        t.nrow_real[l] = P.nrow_real[l];
    }
    hipLaunchKernelGGL(cg_pack_kernel, dim3((unsigned)((t.pre[CN_NLAYER] + 255) / 256)), dim3(256), 0, st, t);
    FCN_CHECK_LAUNCH();
    const int order[CN_NLAYER] = {0, 1, 2, 3, 10, 4, 5, 6, 11, 7, 8, 9, 12, 13};
    for (int q = 0; q < CN_NLAYER; ++q) {
        const int l = order[q];
        CgLayer L;
        cn_fill_layer(d, P, O, ws, feats, one_hot, l, L);
        if (l == 13) { L.y = logits; L.bias = p->bias; L.nbias = P.nrow_real[13]; }
        else if (tr) L.stat = ws->stat + O.st[l];
        const int R = d->B * P.Lout[l];
        {
            const int mt = (R + 63) / 64, ntl = P.N[l] / 64, nch = P.Ktot[l] / KC;
            const int cps = pick_cps(mt * ntl, nch), S = (nch + cps - 1) / cps;
            if (S == 1) {
                hipLaunchKernelGGL(cg_fwd_kernel, dim3(mt, ntl, 1), dim3(CG_T), 0, st, L, (float *)nullptr, nch);
                FCN_CHECK_LAUNCH();
            } else {
                hipLaunchKernelGGL(cg_fwd_kernel, dim3(mt, ntl, S), dim3(CG_T), 0, st, L, ws->partial, cps);
                FCN_CHECK_LAUNCH();
                hipLaunchKernelGGL(cg_fwd_finish_kernel, dim3((R + 127) / 128, ntl), dim3(CG_T), 0, st, L,
                                   (const float *)ws->partial, S);
                FCN_CHECK_LAUNCH();
            }
        }
        if (l != 13) {
            const double M = (double)R * (P.dk[l] > 0 ? P.dk[l] : 1);
            hipLaunchKernelGGL(cn_bn_finalize_kernel, dim3((P.Cs[l] + 63) / 64), dim3(64), 0, st, ws->stat + O.st[l],
                               p->gamma[l], p->beta[l], p->running_mean[l], p->running_var[l],
                               p->num_batches_tracked[l], P.Cs[l], tr, d->eps, d->momentum, M, ws->bn + O.bn[l]);
            FCN_CHECK_LAUNCH();
        }
    }
    return 0;
}

extern "C" int fcn_convnet_backward(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws,
                                    const float *const feats[4], const float *one_hot, const float *dlogits,
                                    float *const dfeats[4], float *const dW[CN_NLAYER], float *const dgamma[CN_NLAYER],
                                    float *const dbeta[CN_NLAYER], float *dbias, void *stream)
{
    if (!d || !p || !ws || !feats || !dlogits || !dfeats || !dW || !dgamma || !dbeta || !dbias) return FCN_E_BADARG;
    if (!d->training) return FCN_E_BADARG;
    if (!ws->y || !ws->dz || !ws->wp || !ws->bn || !ws->bstat || !ws->coef || !ws->partial) return FCN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    CnPlan P;
    FCN_TRY(cn_make_plan(d, P));
    CnOffsets O;
    cn_offsets(d, P, O);
    hipError_t e = hipMemsetAsync(ws->bstat, 0, sizeof(double) * (size_t)O.st[CN_NLAYER], st);
    if (e != hipSuccess) return (int)e;

    // consumers still to come for each producer layer (to know which dgrad is the last one)
    int pending[CN_NLAYER];
    for (int l = 0; l < CN_NLAYER; ++l) pending[l] = 0;
    for (int l = 0; l < CN_NLAYER; ++l)
        for (int s = 0; s < P.nseg[l]; ++s)
            if (P.src[l][s] >= 0) pending[P.src[l][s]] += 1;
    int seen[CN_NLAYER];
    for (int l = 0; l < CN_NLAYER; ++l) seen[l] = 0;

    const int order[CN_NLAYER] = {13, 12, 9, 8, 7, 11, 6, 5, 4, 10, 3, 2, 1, 0};
    for (int q = 0; q < CN_NLAYER; ++q) {
        const int l = order[q];
        CgLayer L;
        cn_fill_layer(d, P, O, ws, feats, one_hot, l, L);
        const int R = d->B * P.Lout[l];
        const float *dz = (l == 13) ? dlogits : ws->dz + O.y[l];
        const float *coef = nullptr;
        if (l == 13) {
            L.y = nullptr;
            hipLaunchKernelGGL(cg_colsum_kernel, dim3(64), dim3(256), 0, st, dlogits, R, 64, P.nrow_real[13], dbias);
            FCN_CHECK_LAUNCH();
        } else {
            const double M = (double)R * (P.dk[l] > 0 ? P.dk[l] : 1);
            hipLaunchKernelGGL(cn_bnbwd_finalize_kernel, dim3((P.Cs[l] + 63) / 64), dim3(64), 0, st, ws->bstat + O.st[l],
                               p->gamma[l], ws->bn + O.bn[l], P.Cs[l], M, ws->coef + O.coef[l], dgamma[l], dbeta[l]);
            FCN_CHECK_LAUNCH();
            coef = ws->coef + O.coef[l];
        }
        // ---- weight gradient
        {
            CgWgrad w;
            w.lay = L; w.dz = dz; w.coef = coef; w.partial = ws->partial;
            w.rows = pick_wrows(R, (P.N[l] / 64) * (P.Ktot[l] / 64));
            const int nsplit = (R + w.rows - 1) / w.rows;
            hipLaunchKernelGGL(cg_wgrad_kernel, dim3(nsplit, P.N[l] / 64, P.Ktot[l] / 64), dim3(CG_T), 0, st, w);
            FCN_CHECK_LAUNCH();
            CgPack pk;
            cn_fill_pack(d, P, l, pk);
            const int64_t ne = (int64_t)P.N[l] * P.Ktot[l];
            hipLaunchKernelGGL(cg_wgrad_reduce_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, ws->partial,
                               nsplit, pk, P.nrow_real[l], dW[l]);
            FCN_CHECK_LAUNCH();
        }
        // ---- data gradients into every non-constant source
        int segoff = 0;
        for (int s = 0; s < P.nseg[l]; ++s) {
            const int src = P.src[l][s];
            if (src != -9) {
                CgDgrad g;
                g.lay = L; g.sg = s; g.segoff = segoff; g.dzc = dz; g.yc = L.y; g.coefc = coef;
                if (src >= 0) {
                    g.ysrc = ws->y + O.y[src]; g.bnsrc = ws->bn + O.bn[src]; g.out = ws->dz + O.y[src];
                    g.accumulate = seen[src] > 0 ? 1 : 0;
                    seen[src] += 1;
                    g.bstat_src = (seen[src] == pending[src]) ? ws->bstat + O.st[src] : nullptr;
                } else {
                    g.ysrc = nullptr; g.bnsrc = nullptr; g.out = dfeats[-src - 1]; g.accumulate = 0; g.bstat_src = nullptr;
                }
                const int Rs = d->B * L.seg[s].Lsrc;
                const int mt = (Rs + 63) / 64, ntl = P.C[l][s] / 64, nch = P.KT[l] * P.N[l] / KC;
                const int cps = pick_cps(mt * ntl, nch), S = (nch + cps - 1) / cps;
                if (S == 1) {
                    hipLaunchKernelGGL(cg_dgrad_kernel, dim3(mt, ntl, 1), dim3(CG_T), 0, st, g, (float *)nullptr, nch);
                    FCN_CHECK_LAUNCH();
                } else {
                    hipLaunchKernelGGL(cg_dgrad_kernel, dim3(mt, ntl, S), dim3(CG_T), 0, st, g, ws->partial, cps);
                    FCN_CHECK_LAUNCH();
                    hipLaunchKernelGGL(cg_dgrad_finish_kernel, dim3((Rs + 127) / 128, ntl), dim3(CG_T), 0, st, g,
                                       (const float *)ws->partial, S);
                    FCN_CHECK_LAUNCH();
                }
            }
            segoff += P.KT[l] * P.C[l][s];
        }
    }
    return 0;
}
