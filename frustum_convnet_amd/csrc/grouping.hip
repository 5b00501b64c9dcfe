// Sliding-frustum grouping (API form) and entry-list compaction for the fused PointNet path.
//
// fcn_query_depth_point_f32 replaces the reference's only native kernel
// (ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-86).  The reference runs one THREAD per
// window that walks all n points serially; here one WAVE owns a window: the frustum's z array is staged
// once per workgroup in LDS, each 64-point chunk is tested in parallel, and __ballot + prefix popcount
// give the ordered compaction ("first nsample hits in index order", padded with the first hit), so the
// int64 rows are written as contiguous coalesced runs.  Results are bit-identical: the predicate is the
// same fp32 fabsf(z_c - z_p) < dis_z.
#include "fcn_common.h"

#define QDP_THREADS 256
#ifndef QDP_WPB
#define QDP_WPB 16                 // windows per workgroup (4 waves x 4 windows)
#endif
#define QDP_LDS_MAX_PTS 16384      // z staged in LDS up to this many points (64 KiB), else read through L1/L2

typedef int64_t __attribute__((address_space(1))) *qdp_grow;
typedef int32_t __attribute__((address_space(1))) *qdp_gcnt;

// The walk of query_depth_point_cuda_kernel.cu:40-64 for the FOUR windows of a wave at once: a 64-point chunk of the z row is read
// once and tested against the four centres (four ballots), so the loop overhead and the LDS read are shared and the four
// compactions interleave.  Per window exactly the reference's sequence: hits in index order, the first `nsample` kept, the rest of the
// row padded with the first hit (zeros for an empty window), cnt = min(hits, nsample).
__device__ __forceinline__ void qdp_scan4(const float *zs, const float *pz, int64_t pt_stride, int use_lds, int n, int m, int mi0,
                                          const float *cz, int64_t ct_stride, float dis_z, int nsample, int64_t *idx_b, int32_t *cnt_b,
                                          int lane)
{
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    float z2[4];
    qdp_grow row[4];
    int c[4], first[4];
    bool live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mi = mi0 + 4 * q;                                // wave-uniform
        live[q] = mi < m;
        z2[q] = live[q] ? cz[(int64_t)mi * ct_stride] : 0.f;
        row[q] = (qdp_grow)(idx_b + (int64_t)(live[q] ? mi : 0) * nsample);
        c[q] = live[q] ? 0 : nsample;
        first[q] = 0;
    }
    for (int k0 = 0; k0 < n; k0 += 64) {
        if (c[0] >= nsample && c[1] >= nsample && c[2] >= nsample && c[3] >= nsample) break;
        const int k = k0 + lane;
        float z1 = 0.f;
        if (k < n) z1 = use_lds ? zs[k] : pz[(int64_t)k * pt_stride];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool hit = (k < n) && c[q] < nsample && (fabsf(z2[q] - z1) < dis_z);
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                if (c[q] == 0) first[q] = k0 + (int)__ffsll((long long)mask) - 1;
                const int pos = c[q] + (int)__popcll(mask & lt_mask);
                if (hit && pos < nsample) row[q][pos] = (int64_t)k;
                c[q] += (int)__popcll(mask);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!live[q]) continue;
        const int taken = c[q] < nsample ? c[q] : nsample;
        const int64_t pad = taken > 0 ? (int64_t)first[q] : (int64_t)0;
        for (int j = taken + lane; j < nsample; j += 64) row[q][j] = pad;
        if (lane == 0) ((qdp_gcnt)cnt_b)[mi0 + 4 * q] = taken;
    }
}

__global__ __launch_bounds__(QDP_THREADS) void qdp_kernel(
    const float *__restrict__ pts_z, int64_t pt_stride, int64_t pt_bstride,
    const float *__restrict__ ctr_z, int64_t ct_stride, int64_t ct_bstride,
    int n, int m, float dis_z, int nsample, int64_t *__restrict__ idx, int32_t *__restrict__ cnt, int use_lds)
{
    FCN_DYN_LDS(float, zs);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.y;
    const float *pz = pts_z + (int64_t)b * pt_bstride;
    if (use_lds) {
        for (int i = tid; i < n; i += QDP_THREADS) zs[i] = pz[(int64_t)i * pt_stride];
        __syncthreads();
    }
    static_assert(QDP_WPB == 16, "a wave takes four windows, a workgroup sixteen");
    qdp_scan4(zs, pz, pt_stride, use_lds, n, m, blockIdx.x * QDP_WPB + wave, ctr_z + (int64_t)b * ct_bstride, ct_stride, dis_z, nsample,
              idx + (int64_t)b * m * nsample, cnt + (int64_t)b * m, lane);
}

extern "C" int fcn_query_depth_point_f32(const float *pts_z, int64_t pt_stride, int64_t pt_bstride,
                                         const float *ctr_z, int64_t ct_stride, int64_t ct_bstride,
                                         int b, int n, int m, float dis_z, int nsample,
                                         int64_t *idx, int32_t *cnt, void *stream)
{
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return FCN_E_BADARG;
    if (b == 0 || m == 0) return 0;
    if (!pts_z && n > 0) return FCN_E_BADARG;
    if (!ctr_z || !idx || !cnt) return FCN_E_BADARG;
    if (b > 65535) return FCN_E_LIMIT;
    const int use_lds = (n <= QDP_LDS_MAX_PTS) ? 1 : 0;
    const size_t lds = use_lds ? (size_t)n * sizeof(float) : 0;
    dim3 grid((m + QDP_WPB - 1) / QDP_WPB, b);
    hipLaunchKernelGGL(qdp_kernel, grid, dim3(QDP_THREADS), lds, (hipStream_t)stream,
                       pts_z, pt_stride, pt_bstride, ctr_z, ct_stride, ct_bstride,
                       n, m, dis_z, nsample, idx, cnt, use_lds);
    FCN_CHECK_LAUNCH();
    return 0;
}

// All scales of a batch in ONE launch: the scales share the point cloud (the z row is staged once per workgroup) and differ in
// window centres, half height and nsample; workgroup x of a frustum takes 16 windows of the scale its index falls into.  The API
// form of one PointNetFeat.forward is 4-5 calls of the operator; each of them is a 14 us affair for ~2 MB of output, most of it
// launch latency.
#define QDP_MAXS 8
struct QdpMulti {
    const float *ctr_z[QDP_MAXS];
    int64_t ct_stride[QDP_MAXS], ct_bstride[QDP_MAXS];
    int64_t *idx[QDP_MAXS];
    int32_t *cnt[QDP_MAXS];
    int m[QDP_MAXS], nsample[QDP_MAXS], blk0[QDP_MAXS + 1];      // blk0: first workgroup (x) of each scale
    float dis_z[QDP_MAXS];
    int nscale;
};

__global__ __launch_bounds__(QDP_THREADS) void qdp_multi_kernel(const float *__restrict__ pts_z, int64_t pt_stride, int64_t pt_bstride,
                                                                int n, QdpMulti a, int use_lds)
{
    FCN_DYN_LDS(float, zs);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const float *pz = pts_z + (int64_t)b * pt_bstride;
    if (use_lds) {
        for (int i = tid; i < n; i += QDP_THREADS) zs[i] = pz[(int64_t)i * pt_stride];
        __syncthreads();
    }
    // the scale of this workgroup: static indices only (a dynamically indexed kernel argument goes through scratch)
    int s = 0;
#pragma unroll
    for (int q = 1; q < QDP_MAXS; ++q)
        if (q < a.nscale && (int)blockIdx.x >= a.blk0[q]) s = q;
    const float *ctr_z = a.ctr_z[0];
    int64_t ct_stride = a.ct_stride[0], ct_bstride = a.ct_bstride[0];
    int64_t *idx = a.idx[0];
    int32_t *cnt = a.cnt[0];
    int m = a.m[0], nsample = a.nsample[0], blk0 = a.blk0[0];
    float dis_z = a.dis_z[0];
#pragma unroll
    for (int q = 1; q < QDP_MAXS; ++q)
        if (s == q) {
            ctr_z = a.ctr_z[q]; ct_stride = a.ct_stride[q]; ct_bstride = a.ct_bstride[q]; idx = a.idx[q]; cnt = a.cnt[q];
            m = a.m[q]; nsample = a.nsample[q]; blk0 = a.blk0[q]; dis_z = a.dis_z[q];
        }
    qdp_scan4(zs, pz, pt_stride, use_lds, n, m, ((int)blockIdx.x - blk0) * QDP_WPB + wave, ctr_z + (int64_t)b * ct_bstride, ct_stride, dis_z,
              nsample, idx + (int64_t)b * m * nsample, cnt + (int64_t)b * m, lane);
}

extern "C" int fcn_query_depth_point_multi_f32(int nscale, const float *pts_z, int64_t pt_stride, int64_t pt_bstride,
                                               const float *const *ctr_z, const int64_t *ct_stride, const int64_t *ct_bstride,
                                               int b, int n, const int32_t *m, const float *dis_z, const int32_t *nsample,
                                               int64_t *const *idx, int32_t *const *cnt, void *stream)
{
    if (nscale < 1 || nscale > QDP_MAXS || b < 0 || n < 0) return FCN_E_BADARG;
    if (!ctr_z || !ct_stride || !ct_bstride || !m || !dis_z || !nsample || !idx || !cnt) return FCN_E_BADARG;
    if (b == 0) return 0;
    if (!pts_z && n > 0) return FCN_E_BADARG;
    if (b > 65535) return FCN_E_LIMIT;
    QdpMulti a;
    a.nscale = nscale;
    int blk = 0;
    for (int s = 0; s < QDP_MAXS; ++s) {
        const int q = s < nscale ? s : 0;
        if (m[q] < 0 || nsample[q] < 0 || (m[q] > 0 && (!ctr_z[q] || !idx[q] || !cnt[q]))) return FCN_E_BADARG;
        a.ctr_z[s] = ctr_z[q]; a.ct_stride[s] = ct_stride[q]; a.ct_bstride[s] = ct_bstride[q]; a.idx[s] = idx[q]; a.cnt[s] = cnt[q];
        a.m[s] = m[q]; a.nsample[s] = nsample[q]; a.dis_z[s] = dis_z[q];
        a.blk0[s] = blk;
        if (s < nscale) blk += (m[q] + QDP_WPB - 1) / QDP_WPB;
    }
    a.blk0[QDP_MAXS] = blk;
    if (blk == 0) return 0;
    const int use_lds = (n <= QDP_LDS_MAX_PTS) ? 1 : 0;
    const size_t lds = use_lds ? (size_t)n * sizeof(float) : 0;
    hipLaunchKernelGGL(qdp_multi_kernel, dim3(blk, b), dim3(QDP_THREADS), lds, (hipStream_t)stream, pts_z, pt_stride, pt_bstride, n, a,
                       use_lds);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Compaction: (idx, cnt) -> entry list.  Window l of frustum b contributes ne = max(cnt,1) rows
// (its distinct hits; an empty window contributes point 0, which the reference also feeds through the
// MLP and the BN statistics before masking, models/det_base.py:95-101).  Row weight = multiplicity in
// the reference's dense (B,C,L,K) tensor: K-ne+1 for the first hit (padding repeats it), 1 otherwise.
// Also accumulates the weighted moments of the centred coordinates u = p - c (fp64), from which the
// conv1 BatchNorm statistics follow exactly (conv1 is linear in u).
// ------------------------------------------------------------------------------------------------
#define CP_THREADS 256
#define CP_SLICES 8

__global__ __launch_bounds__(CP_THREADS) void compact_kernel(
    const float *__restrict__ pc, const float *__restrict__ ref, const int64_t *__restrict__ idx,
    const int32_t *__restrict__ cnt, int N, int L, int K,
    int32_t *__restrict__ woff, float4 *__restrict__ ent, int32_t *__restrict__ ewin, double *__restrict__ mom)
{
    FCN_DYN_LDS(int, offs);        // L+1
    __shared__ int wsum[4];
    __shared__ double red[4][10];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, slice = blockIdx.x;
    const int32_t *cb = cnt + (int64_t)b * L;

    const int per = (L + CP_THREADS - 1) / CP_THREADS;
    const int l0 = min(L, tid * per), l1 = min(L, l0 + per);
    int s = 0;
    for (int l = l0; l < l1; ++l) s += max(cb[l], 1);
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    int run = wbase + incl - s;
    for (int l = l0; l < l1; ++l) {
        offs[l] = run;
        run += max(cb[l], 1);
    }
    if (tid == CP_THREADS - 1) offs[L] = run;
    __syncthreads();
    if (slice == 0)
        for (int l = tid; l <= L; l += CP_THREADS) woff[(int64_t)b * (L + 1) + l] = offs[l];

    const int wps = (L + CP_SLICES - 1) / CP_SLICES;
    const int lbeg = slice * wps, lend = min(L, lbeg + wps);
    const int64_t cap = (int64_t)L * K;
    const float *px = pc + (int64_t)b * 3 * N, *py = px + N, *pzz = py + N;
    const float *rx = ref + (int64_t)b * 3 * L, *ry = rx + L, *rz = ry + L;
    double m[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) m[i] = 0.0;
    for (int l = lbeg + wave; l < lend; l += 4) {
        const int o0 = offs[l], ne = offs[l + 1] - o0;
        const float cx = rx[l], cy = ry[l], cz = rz[l];
        const int64_t *irow = idx + ((int64_t)b * L + l) * K;
        for (int j = lane; j < ne; j += 64) {
            const int p = (int)irow[j];
            const float ux = px[p] - cx, uy = py[p] - cy, uz = pzz[p] - cz;
            const float w = (j == 0) ? (float)(K - ne + 1) : 1.0f;
            const int64_t r = (int64_t)b * cap + o0 + j;
            ent[r] = make_float4(ux, uy, uz, w);
            ewin[r] = l;
            const double dw = w, dx = ux, dy = uy, dz = uz;
            m[0] += dw;
            m[1] += dw * dx; m[2] += dw * dy; m[3] += dw * dz;
            m[4] += dw * dx * dx; m[5] += dw * dx * dy; m[6] += dw * dx * dz;
            m[7] += dw * dy * dy; m[8] += dw * dy * dz; m[9] += dw * dz * dz;
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = wave_sum_f64(m[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < 10) atomic_add_f64(&mom[tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

// Live-tile list: tiles[0] = number of 128-row tiles that hold at least one live row, tiles[4+i] = b*tps + t.
// Every GEMM kernel indexes its row tile through this list, so grids are dense in live work and the wgrad
// splits are balanced no matter how the rows distribute over frustums.
__global__ __launch_bounds__(256) void tile_list_kernel(const int32_t *__restrict__ woff, int B, int L, int tps,
                                                        int32_t *__restrict__ tiles)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + tid;
        const int nt = (b < B) ? (woff[(int64_t)b * (L + 1) + L] + 127) / 128 : 0;
        int incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int off = base + incl - nt;
        for (int t = 0; t < nt; ++t) tiles[4 + off + t] = b * tps + t;
        __syncthreads();
        if (tid == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (tid == 0) tiles[0] = carry_s;
}

extern "C" int fcn_pn_compact(const fcn_pn_desc *d, const float *pc, const float *ref,
                              const int64_t *idx, const int32_t *cnt, const fcn_pn_ws *ws, void *stream)
{
    if (!d || !ws || !pc || !ref || !idx || !cnt) return FCN_E_BADARG;
    if (d->B <= 0 || d->L <= 0 || d->K <= 0 || d->N <= 0) return FCN_E_BADARG;
    if (d->L > 8192 || d->K > 1024 || d->B > 65535) return FCN_E_LIMIT;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws->stat, 0, sizeof(double) * (size_t)(16 + FCN_STAT_REP * (2 * d->C2 + 2 * d->C3)), st);
    if (e != hipSuccess) return (int)e;
    dim3 grid(CP_SLICES, d->B);
    hipLaunchKernelGGL(compact_kernel, grid, dim3(CP_THREADS), sizeof(int) * (size_t)(d->L + 1), st,
                       pc, ref, idx, cnt, d->N, d->L, d->K, ws->woff, (float4 *)ws->ent, ws->ewin,
                       ws->stat + FCN_STAT_MOM);
    FCN_CHECK_LAUNCH();
    const int tps = (d->L * d->K + 127) / 128;
    hipLaunchKernelGGL(tile_list_kernel, dim3(1), dim3(256), 0, st, ws->woff, d->B, d->L, tps, ws->tiles);
    FCN_CHECK_LAUNCH();
    return 0;
}

extern "C" int fcn_arch(void) { return 950; }
