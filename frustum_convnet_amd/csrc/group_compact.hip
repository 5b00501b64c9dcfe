// Fused front of the PointNet scales: sliding-frustum grouping -> entry list, in ONE launch for all scales of a batch.
//
// The API form of the grouping (fcn_query_depth_point_f32, grouping.hip) materialises the reference's int64 idx (B,L,K) --
// 6.9 MB per car batch -- only for fcn_pn_compact to read it back, followed by a one-workgroup tile-list launch, a memset and
// the BN1 finalisation: five graph nodes and ~45 us of dependent latency per scale in front of the first MFMA
// (query_depth_point_cuda_kernel.cu:16-65 + the gather / centre subtract of models/det_base.py:75-80).  Here TWO launches serve
// all scales of a batch:
//   gc_hits_kernel     (chip-wide)  a workgroup takes 16 windows of one (frustum, scale); every wave scans the frustum's z row
//                      (staged in LDS) for its windows: __ballot + prefix popcount give the first K hits in index order -- the
//                      same predicate, order and truncation as the reference kernel -- stored as 16-bit point indices (a
//                      quarter of the int64 idx, never padded, scratch) + the counts;
//   gc_entries_kernel  one workgroup per (frustum, scale): scan of ne = max(cnt, 1) over the windows -> window row offsets;
//                      entry rows (centred coordinates + multiplicity) and window ids, one thread per row; weighted input
//                      moments in fp64.  The LAST frustum of a scale to finish (arrival counter; its 11 inputs travel as
//                      write-through stores, no L2 write-back fence) sums the per-frustum moments in frustum order (bitwise
//                      reproducible), derives the BN1 scale / shift (+ running statistics), builds the live-tile list and
//                      zeroes the BN sum slots of the coming conv launches -- the work of bn1_finalize_kernel,
//                      tile_list_kernel and the memset of fcn_pn_compact.
// (A first version did everything in one launch with a per-frustum arrival counter behind phase A: 2304 agent-scope release
// fences -- each an L2 write-back -- made it 82 us; the kernel boundary is the cheaper hand-off, cdna guide "boundary" row.)
// Outputs are identical to fcn_query_depth_point_f32 + fcn_pn_compact (+ bn1_finalize): cnt, woff, ent, ewin, tiles exactly,
// moments up to fp64 summation order (tests/test_gpu_group_compact.py).
#include "fcn_common.h"
#include "pn_pack.h"

#define GC_T 256
#define GC_WAVES (GC_T / 64)
#ifndef GC_WPB
#define GC_WPB 16                  // windows per workgroup in phase A (4 per wave)
#endif
#define GC_MAX_SCALES 8
#define GC_LDS_MAX_PTS 8192        // z row staged in LDS up to this many points
#define GC_MOM 12                  // doubles per frustum in gmom: 10 moment sums (+ 2 spare)

struct GcScale {
    const float *ref;          // (B,3,L) window centres
    float dis_z;
    int L, K, C1, C2, C3;
    int32_t *woff, *ewin, *tiles, *cnt;
    float4 *ent;
    unsigned short *ghits;     // (B,L,K) first-K hit lists as 16-bit point indices (scratch: lives in ws.y2, free until conv2)
    double *stat;              // 16 + FCN_STAT_REP * (2*C2 + 2*C3)
    double *gmom;              // (B,GC_MOM)
    const float *W1, *gamma, *beta;
    float *rmean, *rvar;
    int64_t *nbt;
    float *bn1;                // scale, shift, mean, rstd (4*C1)
};

struct GcArgs {
    GcScale s[GC_MAX_SCALES];
    PackArgs pk[GC_MAX_SCALES];    // conv2 / conv3 weights of every scale -> ws.wenc (packed by the first launch, on the side)
    const float *pc;           // (B,3,N)
    int B, N, training, use_lds;
    float eps, momentum;
    int fold;                  // 1: the entries launch also folds BN1 (scale / shift, running statistics) from the input moments;
                               // 0: data-only front (fcn_pn_group_compact2 phase 1) -- gc_fold_kernel does it later, after the
                               // optimiser step the weights depend on
};

#define GE_T 1024
#define GE_WAVES (GE_T / 64)

__device__ __forceinline__ GcScale gc_pick(const GcArgs &a, int z)
{
    // the scale is workgroup-uniform: copy its descriptor out of the kernel argument with static indices only
    GcScale S = a.s[0];
#pragma unroll
    for (int q = 1; q < GC_MAX_SCALES; ++q)
        if (z == q) S = a.s[q];
    return S;
}


// BN1 scale / shift (+ running statistics) of one scale from its 10 input moments `mo` (sum w, sum w*u [3], sum w*u*u^T [6]): conv1
// is linear in u, so mean = W1 mu, var = W1^T Cov W1 (fp64).  Called by the last workgroup of gc_entries_kernel (fused front) or by
// gc_fold_kernel (phased front: the moments depend on the batch only, the fold on the weights of THIS step).
__device__ __forceinline__ void gc_fold_bn1(const GcArgs &a, const GcScale &S, const double *mo, int tid, int nthr)
{
    const double M = (double)a.B * (double)S.L * (double)S.K;
    for (int c = tid; c < S.C1; c += nthr) {
        double mean, var;
        if (a.training) {
            // (one fp64 division, no software sqrt: this workgroup is the tail of the front every scale waits for)
            const double iM = 1.0 / M;
            const double mx = mo[1] * iM, my = mo[2] * iM, mz = mo[3] * iM;
            const double cxx = mo[4] * iM - mx * mx, cxy = mo[5] * iM - mx * my, cxz = mo[6] * iM - mx * mz;
            const double cyy = mo[7] * iM - my * my, cyz = mo[8] * iM - my * mz, czz = mo[9] * iM - mz * mz;
            const double w0 = S.W1[3 * c], w1 = S.W1[3 * c + 1], w2 = S.W1[3 * c + 2];
            mean = w0 * mx + w1 * my + w2 * mz;
            var = w0 * (cxx * w0 + cxy * w1 + cxz * w2) + w1 * (cxy * w0 + cyy * w1 + cyz * w2) +
                  w2 * (cxz * w0 + cyz * w1 + czz * w2);
            if (var < 0.0) var = 0.0;
            if (S.rmean) {
                S.rmean[c] = (float)((1.0 - a.momentum) * S.rmean[c] + a.momentum * mean);
                S.rvar[c] = (float)((1.0 - a.momentum) * S.rvar[c] + a.momentum * var * (M / (M - 1.0)));
                if (c == 0 && S.nbt) S.nbt[0] += 1;
            }
        } else {
            mean = S.rmean[c];
            var = S.rvar[c];
        }
        const double rstd = fcn_rsqrt64(var + (double)a.eps);
        const double sc = (double)S.gamma[c] * rstd;
        S.bn1[c] = (float)sc;
        S.bn1[S.C1 + c] = (float)((double)S.beta[c] - mean * sc);
        S.bn1[2 * S.C1 + c] = (float)mean;
        S.bn1[3 * S.C1 + c] = (float)rstd;
    }
}

// ---- launch 1: hit lists (query_depth_point_cuda_kernel.cu:40-64: fabsf(z2 - z1) < dis_z in fp32, ascending k, first K)
__global__ __launch_bounds__(GC_T) void gc_hits_kernel(GcArgs a)
{
    FCN_DYN_LDS(unsigned char, smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const GcScale S = gc_pick(a, (int)blockIdx.z);
    const int L = S.L, K = S.K, N = a.N;
    {   // this scale's weight images, an item or two per thread spread over all (slice, frustum) workgroups of the scale -- the
        // stores are not read before fcn_pn_forward's GEMMs, launches later
        PackArgs P = a.pk[0];
#pragma unroll
        for (int q = 1; q < GC_MAX_SCALES; ++q)
            if ((int)blockIdx.z == q) P = a.pk[q];
        if (P.wenc) pn_pack_range(P, ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * GC_T + tid, (int)(gridDim.x * gridDim.y) * GC_T);
    }
    if ((int)blockIdx.x * GC_WPB >= L) return;
    float *zs = (float *)smem;
    const float *pz = a.pc + (int64_t)b * 3 * N + 2 * (int64_t)N;
    if (a.use_lds) {
        for (int i = tid; i < N; i += GC_T) zs[i] = pz[i];
        __syncthreads();
    }
    const float *rz = S.ref + (int64_t)b * 3 * L + 2 * (int64_t)L;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int q = 0; q < GC_WPB / GC_WAVES; ++q) {
        const int l = blockIdx.x * GC_WPB + q * GC_WAVES + wave;          // wave-uniform
        if (l >= L) break;
        const float z2 = rz[l];
        unsigned short *hrow = S.ghits + ((int64_t)b * L + l) * K;
        int c = 0;
        for (int k0 = 0; k0 < N && c < K; k0 += 64) {
            const int k = k0 + lane;
            float z1 = 0.f;
            if (k < N) z1 = a.use_lds ? zs[k] : pz[k];
            const bool hit = (k < N) && (fabsf(z2 - z1) < S.dis_z);
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                const int pos = c + (int)__popcll(mask & lt_mask);
                if (hit && pos < K) hrow[pos] = (unsigned short)k;
                c += (int)__popcll(mask);
            }
        }
        if (lane == 0) S.cnt[(int64_t)b * L + l] = c < K ? c : K;
    }
}

// ---- launch 2: offsets, entry rows, moments; the last frustum of a scale finalises it
__global__ __launch_bounds__(GE_T) void gc_entries_kernel(GcArgs a)
{
    FCN_DYN_LDS(unsigned char, smem);
    __shared__ int wsum[GE_WAVES];
    __shared__ double red[GE_WAVES][10];
    __shared__ int last_s, carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const GcScale S = gc_pick(a, (int)blockIdx.y);
    const int L = S.L, K = S.K, N = a.N;
    int *cntS = (int *)smem;                   // L
    int *offS = cntS + L;                      // L + 1
    float *cS = (float *)(offS + L + 1);       // 3 * L window centres
    const float *px = a.pc + (int64_t)b * 3 * N, *py = px + N, *pz = py + N;
    const float *rx = S.ref + (int64_t)b * 3 * L;
    for (int l = tid; l < L; l += GE_T) cntS[l] = S.cnt[(int64_t)b * L + l];
    for (int i = tid; i < 3 * L; i += GE_T) cS[i] = rx[i];
    __syncthreads();
    // exclusive scan of ne = max(cnt, 1) over the windows
    {
        const int per = (L + GE_T - 1) / GE_T;
        const int l0 = min(L, tid * per), l1 = min(L, l0 + per);
        int s = 0;
        for (int l = l0; l < l1; ++l) s += max(cntS[l], 1);
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int run = incl - s;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        for (int l = l0; l < l1; ++l) {
            offS[l] = run;
            run += max(cntS[l], 1);
        }
        if (tid == GE_T - 1) offS[L] = run;
        __syncthreads();
        for (int l = tid; l < L; l += GE_T) S.woff[(int64_t)b * (L + 1) + l] = offS[l];
        // woff[b][L] (the frustum's live-row count) is read by the finalising workgroup: write-through
        if (tid == 0) __hip_atomic_store(&S.woff[(int64_t)b * (L + 1) + L], offS[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // one thread per entry row (window by binary search in the offsets) + weighted moments of u = p - c
    const int64_t cap = (int64_t)L * K;
    const int nent = offS[L];
    double m[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) m[i] = 0.0;
    for (int e = tid; e < nent; e += GE_T) {
        int lo = 0, hi = L;                                               // largest l with offS[l] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offS[mid] <= e) lo = mid; else hi = mid;
        }
        const int l = lo, j = e - offS[l], ne = offS[l + 1] - offS[l];
        // empty window: point 0 (the reference's zero-initialised idx row)
        const int p = (cntS[l] > 0) ? (int)S.ghits[((int64_t)b * L + l) * K + j] : 0;
        const float ux = px[p] - cS[l], uy = py[p] - cS[L + l], uz = pz[p] - cS[2 * L + l];
        const float w = (j == 0) ? (float)(K - ne + 1) : 1.0f;
        const int64_t r = (int64_t)b * cap + e;
        S.ent[r] = make_float4(ux, uy, uz, w);
        S.ewin[r] = l;
        const double dw = w, dx = ux, dy = uy, dz = uz;
        m[0] += dw;
        m[1] += dw * dx; m[2] += dw * dy; m[3] += dw * dz;
        m[4] += dw * dx * dx; m[5] += dw * dx * dy; m[6] += dw * dx * dz;
        m[7] += dw * dy * dy; m[8] += dw * dy * dz; m[9] += dw * dz * dz;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double v = wave_sum_f64(m[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        // the 10 moment sums leave as write-through (agent-scope) stores; drained, then the arrival counter -- the
        // finalising workgroup reads them with agent-scope loads: no L2 write-back fence on either side (cdna guide G16, R1)
        for (int i = 0; i < 10; ++i) {
            double v = 0.0;
            for (int w = 0; w < GE_WAVES; ++w) v += red[w][i];
            __hip_atomic_store(&S.gmom[(int64_t)b * GC_MOM + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = atomicAdd(&S.tiles[1], 1);                     // tiles[1]: arrival counter, 0 between launches
        last_s = (ticket == a.B - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last_s) return;

    // ======== the last frustum of this scale
    // moments in frustum order (fixed order: reproducible bit for bit).  All B x 10 values are fetched in parallel into LDS
    // first -- a serial chain of B agent-scope loads cost ~20 us here
    {
        double *gm = (double *)smem;                                       // (cntS / offS / cS are dead by now)
        const int nv = a.B * GC_MOM;
        const int cap = (int)(((size_t)(2 * L + 1) * sizeof(int) + (size_t)3 * L * sizeof(float)) / sizeof(double));
        const bool fits = nv <= cap;
        __syncthreads();
        if (fits)
            for (int i = tid; i < nv; i += GE_T) gm[i] = __hip_atomic_load(&S.gmom[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid < 10) {
            double v = 0.0;
            for (int q = 0; q < a.B; ++q)
                v += fits ? gm[q * GC_MOM + tid]
                          : __hip_atomic_load(&S.gmom[(int64_t)q * GC_MOM + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.stat[FCN_STAT_MOM + tid] = v;
            red[0][tid] = v;
        }
    }
    // BN sum slots of the conv launches that follow
    for (int i = 10 + tid; i < 16 + FCN_STAT_REP * (2 * S.C2 + 2 * S.C3); i += GE_T) S.stat[i] = 0.0;      // every replica block
    __syncthreads();
    // BN1 scale / shift from the moments (conv1 is linear in u): what bn1_finalize_kernel computes
    if (a.fold) gc_fold_bn1(a, S, &red[0][0], tid, GE_T);
    // live-tile list: tiles[0] = count, tiles[4+i] = b*tps + t (frustum-major, as tile_list_kernel)
    {
        const int tps = (int)((cap + 127) / 128);
        if (tid == 0) carry_s = 0;
        __syncthreads();
        for (int b0 = 0; b0 < a.B; b0 += GE_T) {
            const int bb = b0 + tid;
            int ne = 0;
            if (bb < a.B) ne = __hip_atomic_load(&S.woff[(int64_t)bb * (L + 1) + L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int nt = (ne + 127) / 128;
            int incl = nt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = carry_s;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            const int off = base + incl - nt;
            for (int t = 0; t < nt; ++t) S.tiles[4 + off + t] = bb * tps + t;
            __syncthreads();
            if (tid == GE_T - 1) carry_s = base + incl;
            __syncthreads();
        }
        if (tid == 0) {
            S.tiles[0] = carry_s;
            S.tiles[1] = 0;                                               // counter ready for the next launch
        }
    }
}


// ---- phased front, phase 2: everything of the front that depends on the WEIGHTS (they change with every optimiser step) -- the
// split-encoded conv2 / conv3 images of every scale and the BN1 fold -- in one light launch behind the optimiser step; phase 1
// (hit lists, entry rows, moments, tile lists: functions of the batch alone) may have run long before, beside the previous step
#define GF_T 256
__global__ __launch_bounds__(GF_T) void gc_fold_kernel(GcArgs a)
{
    __shared__ double mo[10];
    const int tid = threadIdx.x;
    const GcScale S = gc_pick(a, (int)blockIdx.y);
    {   // ONE image item (8 reduction-adjacent values of a weight row or column -> both planes) per thread: the grid covers the
        // widest scale's item count, a narrower scale's surplus workgroups leave at once.  (16 workgroups per scale looping over
        // the items took 14 us -- a dozen dependent memory round trips per thread -- on the chain behind the optimiser step.)
        PackArgs P = a.pk[0];
#pragma unroll
        for (int q = 1; q < GC_MAX_SCALES; ++q)
            if ((int)blockIdx.y == q) P = a.pk[q];
        if (P.wenc) pn_pack_range(P, (int)blockIdx.x * GF_T + tid, (int)gridDim.x * GF_T);
    }
    if (blockIdx.x != 0) return;
    if (tid < 10) mo[tid] = S.stat[FCN_STAT_MOM + tid];
    __syncthreads();
    gc_fold_bn1(a, S, mo, tid, GF_T);
}

extern "C" int fcn_pn_group_compact2(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p, const float *pc,
                                     const float *const *ref, const float *dis_z, const fcn_pn_ws *const *ws,
                                     int32_t *const *cnt, int phase, void *stream);

extern "C" int fcn_pn_group_compact(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p, const float *pc,
                                    const float *const *ref, const float *dis_z, const fcn_pn_ws *const *ws,
                                    int32_t *const *cnt, void *stream)
{
    return fcn_pn_group_compact2(nscale, d, p, pc, ref, dis_z, ws, cnt, 3, stream);
}

// phase 1: the batch-only part (hit lists, entries, moments, tile lists, zeroed BN sums) -- a loader may run it for the NEXT batch
// beside the current step; phase 2: the weight-dependent part (weight images + BN1 fold), one light launch; 3: both, fused (the
// two launches of fcn_pn_group_compact).  Phases 1 + 2 leave the workspaces exactly as phase 3 does.
extern "C" int fcn_pn_group_compact2(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p, const float *pc,
                                     const float *const *ref, const float *dis_z, const fcn_pn_ws *const *ws,
                                     int32_t *const *cnt, int phase, void *stream)
{
    if (phase < 1 || phase > 3) return FCN_E_BADARG;
    if (nscale < 1 || nscale > GC_MAX_SCALES || !d || !p || !pc || !ref || !dis_z || !ws || !cnt) return FCN_E_BADARG;
    GcArgs a;
    size_t lds = 0;
    int maxslice = 1;
    for (int s = 0; s < GC_MAX_SCALES; ++s) {
        const int q = s < nscale ? s : 0;
        if (!d[q] || !p[q] || !ref[q] || !ws[q] || !cnt[q]) return FCN_E_BADARG;
        const fcn_pn_desc *D = d[q];
        if (D->B != d[0]->B || D->N != d[0]->N || D->training != d[0]->training || D->eps != d[0]->eps ||
            D->momentum != d[0]->momentum)
            return FCN_E_BADARG;
        if (D->B <= 0 || D->L <= 0 || D->K <= 0 || D->N <= 0) return FCN_E_BADARG;
        if (D->N > 65535 || D->L > 8192 || D->K > 1024 || D->B > 65535) return FCN_E_LIMIT;       // 16-bit point indices
        if (D->C1 % 64 || D->C2 % 64 || D->C3 % 64) return FCN_E_BADARG;
        if (!ws[q]->woff || !ws[q]->ent || !ws[q]->ewin || !ws[q]->tiles || !ws[q]->stat || !ws[q]->bn || !ws[q]->gmom ||
            !ws[q]->y2 || !ws[q]->wenc)
            return FCN_E_BADARG;
        GcScale &S = a.s[s];
        S.ref = ref[q]; S.dis_z = dis_z[q]; S.L = D->L; S.K = D->K; S.C1 = D->C1; S.C2 = D->C2; S.C3 = D->C3;
        S.woff = ws[q]->woff; S.ewin = ws[q]->ewin; S.tiles = ws[q]->tiles; S.cnt = cnt[q]; S.ent = (float4 *)ws[q]->ent;
        S.ghits = (unsigned short *)ws[q]->y2;          // (B,L,K) x 2 bytes <= (B,L*K,C2) x 4 bytes; y2 is written by conv2 later
        S.stat = ws[q]->stat; S.gmom = ws[q]->gmom;
        S.W1 = p[q]->W[0]; S.gamma = p[q]->gamma[0]; S.beta = p[q]->beta[0]; S.rmean = p[q]->running_mean[0];
        S.rvar = p[q]->running_var[0]; S.nbt = p[q]->num_batches_tracked[0];
        S.bn1 = ws[q]->bn + fcn_bn_off(0, D->C1, D->C2);
        PackArgs &K = a.pk[s];
        K.W2 = p[q]->W[1]; K.W3 = p[q]->W[2]; K.wenc = (s < nscale && phase != 1) ? ws[q]->wenc : nullptr;
        K.C1 = D->C1; K.C2 = D->C2; K.C3 = D->C3; K.precision = D->precision;
        if (!K.W2 || !K.W3 || ((uintptr_t)ws[q]->wenc & 15) || D->precision < 0 || D->precision > FCN_PREC_BF16_OPS) return FCN_E_BADARG;
        if (!D->training && (!S.rmean || !S.rvar)) return FCN_E_BADARG;
        const size_t need = (size_t)(2 * D->L + 1) * sizeof(int) + (size_t)3 * D->L * sizeof(float);
        if (need > lds) lds = need;
        const int ns_ = (D->L + GC_WPB - 1) / GC_WPB;
        if (s < nscale && ns_ > maxslice) maxslice = ns_;
    }
    a.pc = pc; a.B = d[0]->B; a.N = d[0]->N; a.training = d[0]->training ? 1 : 0; a.eps = d[0]->eps; a.momentum = d[0]->momentum;
    a.use_lds = (a.N <= GC_LDS_MAX_PTS) ? 1 : 0;
    a.fold = (phase == 3) ? 1 : 0;
    if (lds > 64 * 1024) return FCN_E_LIMIT;
    hipStream_t st = (hipStream_t)stream;
    if (phase == 2) {
        int items = 1;              // image items of the widest scale: 2 * (C2*C1 + C3*C2) / 8 (49 k for 256-256-512)
        for (int s = 0; s < nscale; ++s) {
            const int n = 2 * (d[s]->C2 * d[s]->C1 + d[s]->C3 * d[s]->C2) / 8;
            if (n > items) items = n;
        }
        hipLaunchKernelGGL(gc_fold_kernel, dim3((items + GF_T - 1) / GF_T, nscale), dim3(GF_T), 0, st, a);
        FCN_CHECK_LAUNCH();
        return 0;
    }
    // (the split-encoded conv2 / conv3 weights of every scale -- read by the GEMMs of fcn_pn_forward / fcn_pn_backward -- are
    // packed by gc_hits_kernel on the side)
    hipLaunchKernelGGL(gc_hits_kernel, dim3(maxslice, a.B, nscale), dim3(GC_T), a.use_lds ? (size_t)a.N * sizeof(float) : 0, st, a);
    FCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gc_entries_kernel, dim3(a.B, nscale), dim3(GE_T), lds, st, a);
    FCN_CHECK_LAUNCH();
    return 0;
}
