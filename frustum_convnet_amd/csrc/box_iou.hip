// Rotated-box kernels after the heads:
//   iou_metric_kernel   the IoU training metrics of models/det_base.py:480-503 (IoU_2D, IoU_3D, IoU_>=thresh) from the logits
//   iou_pair_kernel     rbbox_iou_3d_pair, ops/pybind11/box_ops.h:173-260 (call site models/det_base.py:495): paired BEV / 3-D
//                       IoU of two corner arrays
//   decode_kernel       the per-frustum numpy loop of train/test_net_det.py:254-293: foreground selection (p_bg < p_fg, or the
//                       arg-max of p_fg when a frustum has none / cfg.TEST.METHOD == 'top'), arg-max heading bin / size
//                       cluster decode (models/box_transform.py:28-41,5-12), from_prediction_to_label_format
//                       (datasets/provider_sample.py:375-387), the h/w/l >= 0.01 filter and the score p_fg + rgb_prob
//   nms_kernel          rotate_nms_3d_cc (ops/pybind11/rbbox_iou.py:294-311) -> rotate_non_max_suppression_3d_cpu
//                       (ops/pybind11/nms_cpu.h:148-240): per (frame, class) group, greedy in descending score order,
//                       suppress when the rotated 3-D IoU >= thresh, keep the first top_k
// The reference leaves the device for all three (numpy loops + boost::geometry polygon clipping on the host); here the
// detections never leave HBM until the final keep lists are read.  The clip core is box_iou.h.
#include "fcn_common.h"
#define FCN_HD __device__ __forceinline__
#include "box_iou.h"

// ------------------------------------------------------------------------------------------------
__global__ void iou_pair_kernel(const float *__restrict__ c1, const float *__restrict__ c2, int n, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = c1 + (int64_t)i * 24, *b = c2 + (int64_t)i * 24;
    // BEV polygon = corners 6,7,4,5 (x,z); y extents from corners 0 and 4 (box_ops.h:208-232)
    const int ord[4] = {6, 7, 4, 5};
    float ax[4], az[4], bx[4], bz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ax[k] = a[ord[k] * 3]; az[k] = a[ord[k] * 3 + 2];
        bx[k] = b[ord[k] * 3]; bz[k] = b[ord[k] * 3 + 2];
    }
    float i2, i3;
    fcn_iou_from_polys(ax, az, a[1], a[13], bx, bz, b[1], b[13], &i2, &i3);
    out[2 * i] = i2;
    out[2 * i + 1] = i3;
}

extern "C" int fcn_box3d_iou_pair_f32(const float *corners1, const float *corners2, int n, float *out2, void *stream)
{
    if (n < 0 || (n > 0 && (!corners1 || !corners2 || !out2))) return FCN_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(iou_pair_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, corners1, corners2, n, out2);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// IoU training metrics (models/det_base.py:480-503): on every foreground row (cls_label == 1) the box decoded with the
// ARG-MAX heading bin / size cluster against the label box -> means of the BEV IoU, the 3-D IoU and [3-D IoU >= thresh].
// The reference copies both corner arrays to the host every step and clips with boost there; here it is one small launch
// that the binding puts on a side stream BESIDE the loss tail (nothing in the backward depends on it), so the step's
// critical path does not see it.  One thread per row; the few foreground lanes do the clipping; sums go to a persistent
// scratch (zero between launches) and the last workgroup (arrival counter) writes the three means + the foreground count.
#define IOM_T 128

struct IouMetricArgs {
    const float *logits;       // (B*L2, ld): cols 0..1 cls, 2.. reg
    const int64_t *cls_label;  // (B,L2)
    const float *ref2;         // (B,3,L2)
    const float *box_center, *box_heading, *box_size;      // (B,3) (B,1) (B,3)
    const float *mean_size;    // (ns,3)
    float *scratch;            // 8 floats: [0..2] sums, [3] fg count, [4] arrival counter (as int); zero between launches
    float *out4;               // IoU_2D, IoU_3D, IoU_>=thresh, nfg
    int B, L2, ld, nb, ns;
    float thresh;
};

__global__ __launch_bounds__(IOM_T) void iou_metric_kernel(IouMetricArgs a)
{
    __shared__ float sh[4][IOM_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.B * a.L2, nb = a.nb, ns = a.ns;
    const int r = blockIdx.x * IOM_T + tid;
    float s2 = 0.f, s3 = 0.f, st = 0.f, sn = 0.f;
    if (r < R && a.cls_label[r] == 1) {
        const int b = r / a.L2, l = r % a.L2;
        const float *o = a.logits + (int64_t)r * a.ld + 2;
        const float per = 6.283185307179586f / (float)nb, half = per * 0.5f;
        int ah = 0, as = 0;
        float mx = o[3];
        for (int j = 1; j < nb; ++j)
            if (o[3 + j] > mx) { mx = o[3 + j]; ah = j; }
        const float *ss = o + 3 + 2 * nb;
        mx = ss[0];
        for (int j = 1; j < ns; ++j)
            if (ss[j] > mx) { mx = ss[j]; as = j; }
        float pa = (float)ah * per + o[3 + nb + ah] * half;
        if (pa > 3.141592653589793f) pa -= 6.283185307179586f;
        const float *sr3 = o + 3 + 2 * nb + ns + 3 * as;
        const float m0 = a.mean_size[as * 3], m1 = a.mean_size[as * 3 + 1], m2 = a.mean_size[as * 3 + 2];
        const float pcx = o[0] + a.ref2[((int64_t)b * 3 + 0) * a.L2 + l], pcy = o[1] + a.ref2[((int64_t)b * 3 + 1) * a.L2 + l],
                    pcz = o[2] + a.ref2[((int64_t)b * 3 + 2) * a.L2 + l];
        const float hl = a.box_heading[b];
        float i2, i3;
        fcn_iou_from_params(pcx, pcy, pcz, sr3[0] * m0 + m0, sr3[1] * m1 + m1, sr3[2] * m2 + m2, cosf(pa), sinf(pa),
                            a.box_center[b * 3], a.box_center[b * 3 + 1], a.box_center[b * 3 + 2], a.box_size[b * 3],
                            a.box_size[b * 3 + 1], a.box_size[b * 3 + 2], cosf(hl), sinf(hl), &i2, &i3);
        s2 = i2; s3 = i3; st = (i3 >= a.thresh) ? 1.f : 0.f; sn = 1.f;
    }
    s2 = wave_sum_f32(s2); s3 = wave_sum_f32(s3); st = wave_sum_f32(st); sn = wave_sum_f32(sn);
    if (lane == 0) { sh[0][wave] = s2; sh[1][wave] = s3; sh[2][wave] = st; sh[3][wave] = sn; }
    __syncthreads();
    if (tid == 0) {
        float t[4];
        for (int q = 0; q < 4; ++q) {
            t[q] = 0.f;
            for (int w = 0; w < IOM_T / 64; ++w) t[q] += sh[q][w];
            if (t[q] != 0.f) atomicAdd(&a.scratch[q], t[q]);
        }
        __threadfence();
        const int ticket = atomicAdd((int *)&a.scratch[4], 1);
        if (ticket == (int)gridDim.x - 1) {
            __threadfence();
            float v[4];
            for (int q = 0; q < 4; ++q) v[q] = __hip_atomic_load(&a.scratch[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float inv = v[3] > 0.f ? 1.f / v[3] : 0.f;
            a.out4[0] = v[0] * inv; a.out4[1] = v[1] * inv; a.out4[2] = v[2] * inv; a.out4[3] = v[3];
            for (int q = 0; q < 4; ++q) a.scratch[q] = 0.f;
            ((int *)a.scratch)[4] = 0;
        }
    }
}

extern "C" int fcn_det_iou_metrics(const float *logits, int ld, const int64_t *cls_label, const float *center_ref2,
                                   const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                                   const float *mean_size, int B, int L2, int num_heading_bin, int num_size_cluster,
                                   float iou_thresh, float *scratch8, float *out4, void *stream)
{
    if (!logits || !cls_label || !center_ref2 || !box3d_center || !box3d_heading || !box3d_size || !mean_size || !scratch8 ||
        !out4)
        return FCN_E_BADARG;
    if (B <= 0 || L2 <= 0 || num_heading_bin < 1 || num_size_cluster < 1) return FCN_E_BADARG;
    if (ld < 2 + 3 + 2 * num_heading_bin + 4 * num_size_cluster) return FCN_E_BADARG;
    IouMetricArgs a;
    a.logits = logits; a.cls_label = cls_label; a.ref2 = center_ref2; a.box_center = box3d_center;
    a.box_heading = box3d_heading; a.box_size = box3d_size; a.mean_size = mean_size; a.scratch = scratch8; a.out4 = out4;
    a.B = B; a.L2 = L2; a.ld = ld; a.nb = num_heading_bin; a.ns = num_size_cluster; a.thresh = iou_thresh;
    hipLaunchKernelGGL(iou_metric_kernel, dim3((B * L2 + IOM_T - 1) / IOM_T), dim3(IOM_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
#define DEC_T 256
#define DEC_NB_MAX 16
#define DEC_NS_MAX 8

struct DecodeArgs {
    const float *logits;       // (B*L2, ld) rows: cols 0..1 cls, 2.. reg (3 centre, nb heading scores, nb residuals, ns size scores, 3*ns)
    const float *ref2;         // (B,3,L2) centres of the output positions
    const float *mean_size;    // (ns,3)
    const float *rot_angle;    // (B)
    const float *ref_center;   // (B,3) or nullptr (zeros: not the refinement stage)
    const float *rgb_prob;     // (B) or nullptr (ones: boxes from ground truth 2-D detections)
    float *dets;               // (B*L2, 8): tx, ty, tz, l, w, h, ry, score   (the order rotate_nms_3d_cc takes)
    int32_t *valid;            // (B*L2)
    int B, L2, ld, nb, ns, method;
};

__global__ __launch_bounds__(DEC_T) void decode_kernel(DecodeArgs a)
{
    __shared__ float best_s[DEC_T];
    __shared__ int best_i[DEC_T];
    __shared__ int nfg_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L2 = a.L2, nb = a.nb, ns = a.ns;
    if (tid == 0) nfg_s = 0;
    __syncthreads();
    // pass 1: foreground count and the arg-max of p_fg (first maximum, like np.argmax)
    float bs = -1.f;
    int bi = 0x7fffffff, cnt = 0;
    for (int l = tid; l < L2; l += DEC_T) {
        const float *row = a.logits + ((int64_t)b * L2 + l) * a.ld;
        const float c0 = row[0], c1 = row[1];
        const float m = fmaxf(c0, c1), e0 = expf(c0 - m), e1 = expf(c1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        cnt += (p0 < p1) ? 1 : 0;
        if (p1 > bs) { bs = p1; bi = l; }
    }
    best_s[tid] = bs;
    best_i[tid] = bi;
    if (cnt) atomicAdd(&nfg_s, cnt);
    __syncthreads();
    for (int o = DEC_T / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const float s2 = best_s[tid + o];
            const int i2 = best_i[tid + o];
            if (s2 > best_s[tid] || (s2 == best_s[tid] && i2 < best_i[tid])) { best_s[tid] = s2; best_i[tid] = i2; }
        }
        __syncthreads();
    }
    const bool use_top = (a.method == 0) || (nfg_s == 0);
    const int top = best_i[0];
    const float rot = a.rot_angle[b];
    const float cr = cosf(-rot), sr = sinf(-rot);
    const float rcx = a.ref_center ? a.ref_center[b * 3] : 0.f, rcy = a.ref_center ? a.ref_center[b * 3 + 1] : 0.f,
                rcz = a.ref_center ? a.ref_center[b * 3 + 2] : 0.f;
    const float rgb = a.rgb_prob ? a.rgb_prob[b] : 1.f;
    const float per = 6.283185307179586f / (float)nb, half = per * 0.5f;
    // pass 2: decode every selected position
    for (int l = tid; l < L2; l += DEC_T) {
        const int64_t r = (int64_t)b * L2 + l;
        const float *row = a.logits + r * a.ld;
        const float c0 = row[0], c1 = row[1];
        const float m = fmaxf(c0, c1), e0 = expf(c0 - m), e1 = expf(c1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const bool sel = use_top ? (l == top) : (p0 < p1);
        const float *o = row + 2;
        int ah = 0, as = 0;
        float mx = o[3];
        for (int j = 1; j < nb; ++j)
            if (o[3 + j] > mx) { mx = o[3 + j]; ah = j; }
        const float *ss = o + 3 + 2 * nb;
        mx = ss[0];
        for (int j = 1; j < ns; ++j)
            if (ss[j] > mx) { mx = ss[j]; as = j; }
        float ang = (float)ah * per + o[3 + nb + ah] * half;             // angle_decode
        if (ang > 3.141592653589793f) ang -= 6.283185307179586f;
        const float *sr3 = o + 3 + 2 * nb + ns + 3 * as;                  // size_decode
        const float sl = sr3[0] * a.mean_size[as * 3] + a.mean_size[as * 3];
        const float sw = sr3[1] * a.mean_size[as * 3 + 1] + a.mean_size[as * 3 + 1];
        const float sh = sr3[2] * a.mean_size[as * 3 + 2] + a.mean_size[as * 3 + 2];
        const float cx = o[0] + a.ref2[((int64_t)b * 3 + 0) * L2 + l], cy = o[1] + a.ref2[((int64_t)b * 3 + 1) * L2 + l],
                    cz = o[2] + a.ref2[((int64_t)b * 3 + 2) * L2 + l];
        // from_prediction_to_label_format: rotate_pc_along_y(centre, -rot) (+ ref_center), bottom centre, ry = heading + rot
        const float tx = cr * cx - sr * cz + rcx;
        const float tz = sr * cx + cr * cz + rcz;
        const float ty = cy + rcy + 0.5f * sh;
        float *d = a.dets + r * 8;
        d[0] = tx; d[1] = ty; d[2] = tz; d[3] = sl; d[4] = sw; d[5] = sh; d[6] = ang + rot; d[7] = p1 + rgb;
        a.valid[r] = (sel && !(sh < 0.01f || sw < 0.01f || sl < 0.01f)) ? 1 : 0;
    }
}

extern "C" int fcn_decode_detections(const float *logits, int ld, const float *center_ref2, const float *mean_size,
                                     const float *rot_angle, const float *ref_center, const float *rgb_prob, int B, int L2,
                                     int num_heading_bin, int num_size_cluster, int method, float *dets, int32_t *valid,
                                     void *stream)
{
    if (!logits || !center_ref2 || !mean_size || !rot_angle || !dets || !valid) return FCN_E_BADARG;
    if (B <= 0 || L2 <= 0 || num_heading_bin < 1 || num_size_cluster < 1) return FCN_E_BADARG;
    if (ld < 2 + 3 + 2 * num_heading_bin + 4 * num_size_cluster) return FCN_E_BADARG;
    if (method != 0 && method != 1) return FCN_E_BADARG;
    DecodeArgs a;
    a.logits = logits; a.ref2 = center_ref2; a.mean_size = mean_size; a.rot_angle = rot_angle; a.ref_center = ref_center;
    a.rgb_prob = rgb_prob; a.dets = dets; a.valid = valid; a.B = B; a.L2 = L2; a.ld = ld; a.nb = num_heading_bin;
    a.ns = num_size_cluster; a.method = method;
    hipLaunchKernelGGL(decode_kernel, dim3(B), dim3(DEC_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
#define NMS_T 256
#define NMS_MAXC 4096          // candidates of one (frame, class) group held in LDS

struct NmsArgs {
    const float *dets;         // (n rows, 8): tx, ty, tz, l, w, h, ry, score (label format, as rotate_nms_3d_cc receives it)
    const int32_t *valid;      // (n rows) or nullptr (all valid)
    const int32_t *unit_group; // (U): group of each unit (a unit = `rows_per_unit` consecutive rows, one frustum)
    int U, rows_per_unit, G, top_k;
    float thresh;
    int32_t *keep;             // (G, top_k) row indices in keep order
    int32_t *keep_cnt;         // (G): kept count, or -1 when the group had more than NMS_MAXC candidates
};

// geometry of one detection row for the overlap test.  The reference feeds (tx, ty, tz, l, w, h, ry) straight into
// boxes3d2corners, i.e. it treats ty (the bottom-centre y of the label format) as the box centre's y for BOTH boxes of a
// pair; the y overlap only depends on differences, so this restates it literally.
struct NmsBox {
    float x[4], z[4], ytop, ybot, lo[3], hi[3];
};

__device__ __forceinline__ void nms_load(const float *d, NmsBox &b)
{
    const float co = cosf(d[6]), si = sinf(d[6]);
    fcn_bev_rect(d[0], d[2], d[3], d[4], co, si, b.x, b.z);
    b.ytop = d[1] + 0.5f * d[5];
    b.ybot = d[1] - 0.5f * d[5];
    b.lo[0] = fminf(fminf(b.x[0], b.x[1]), fminf(b.x[2], b.x[3]));
    b.hi[0] = fmaxf(fmaxf(b.x[0], b.x[1]), fmaxf(b.x[2], b.x[3]));
    b.lo[2] = fminf(fminf(b.z[0], b.z[1]), fminf(b.z[2], b.z[3]));
    b.hi[2] = fmaxf(fmaxf(b.z[0], b.z[1]), fmaxf(b.z[2], b.z[3]));
    b.lo[1] = b.ybot;
    b.hi[1] = b.ytop;
}

__global__ __launch_bounds__(NMS_T) void nms_kernel(NmsArgs a)
{
    __shared__ float key[NMS_MAXC];            // scores (sorted descending)
    __shared__ int32_t row[NMS_MAXC];          // row index of each candidate
    __shared__ unsigned char dead[NMS_MAXC];
    __shared__ int n_s, cur_s, kept_s;
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { n_s = 0; kept_s = 0; }
    __syncthreads();
    // ---- gather the group's candidates in row order (units in order, rows in order: the reference's append order)
    for (int u = 0; u < a.U; ++u) {
        if (a.unit_group[u] != g) continue;                              // workgroup-uniform
        for (int base = 0; base < a.rows_per_unit; base += NMS_T) {
            const int l = base + tid;
            const int r = u * a.rows_per_unit + l;
            const bool ok = l < a.rows_per_unit && (!a.valid || a.valid[r] != 0);
            // ordered append: ballot / prefix within the wave, then across the 4 waves through LDS counters
            const unsigned long long mask = __ballot(ok);
            const int lane = tid & 63, wave = tid >> 6;
            __shared__ int wcnt[NMS_T / 64];
            if (lane == 0) wcnt[wave] = __popcll(mask);
            __syncthreads();
            int off = n_s;
            for (int w = 0; w < wave; ++w) off += wcnt[w];
            const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
            if (ok && pos < NMS_MAXC) { key[pos] = a.dets[(int64_t)r * 8 + 7]; row[pos] = r; }
            __syncthreads();
            if (tid == 0) {
                int t = 0;
                for (int w = 0; w < NMS_T / 64; ++w) t += wcnt[w];
                n_s += t;
            }
            __syncthreads();
        }
    }
    const int n = n_s;
    if (n > NMS_MAXC) {
        if (tid == 0) a.keep_cnt[g] = -1;
        return;
    }
    // ---- bitonic sort, descending by score (ties: larger row first, the order of np.argsort(...)[::-1] on a stable sort)
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += NMS_T) { key[i] = -INFINITY; row[i] = -1; }
    for (int i = tid; i < np2; i += NMS_T) dead[i] = 0;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NMS_T) {
                const int p = i ^ j;
                if (p > i) {
                    const bool desc = (i & k) == 0;
                    const float ki = key[i], kp = key[p];
                    const int ri = row[i], rp = row[p];
                    const bool i_first = (ki > kp) || (ki == kp && ri > rp);       // i belongs before p in descending order
                    if (desc ? !i_first : i_first) { key[i] = kp; key[p] = ki; row[i] = rp; row[p] = ri; }
                }
            }
            __syncthreads();
        }
    }
    // ---- greedy suppression: sequential over the kept boxes, parallel over the remaining candidates
    int i = 0;
    while (true) {
        if (tid == 0) {
            int c = i;
            while (c < n && dead[c]) ++c;
            cur_s = c;
            if (c < n) {
                if (kept_s < a.top_k) a.keep[(int64_t)g * a.top_k + kept_s] = row[c];
                kept_s += 1;
            }
        }
        __syncthreads();
        const int c = cur_s;
        if (c >= n) break;
        NmsBox bi;
        nms_load(a.dets + (int64_t)row[c] * 8, bi);
        for (int j = c + 1 + tid; j < n; j += NMS_T) {
            if (dead[j]) continue;
            NmsBox bj;
            nms_load(a.dets + (int64_t)row[j] * 8, bj);
            // standup_iou <= 0: the axis-aligned hulls do not overlap in some dimension -> skipped (nms_cpu.h:195)
            bool apart = false;
#pragma unroll
            for (int q = 0; q < 3; ++q) apart = apart || (fminf(bi.hi[q], bj.hi[q]) - fmaxf(bi.lo[q], bj.lo[q]) <= 0.f);
            if (apart) continue;
            float i2, i3;
            fcn_iou_from_polys(bi.x, bi.z, bi.ytop, bi.ybot, bj.x, bj.z, bj.ytop, bj.ybot, &i2, &i3);
            if (i3 >= a.thresh) dead[j] = 1;
        }
        i = c + 1;
        __syncthreads();
    }
    if (tid == 0) a.keep_cnt[g] = kept_s < a.top_k ? kept_s : a.top_k;
}

extern "C" int fcn_rotate_nms_3d(const float *dets, const int32_t *valid, const int32_t *unit_group, int num_units,
                                 int rows_per_unit, int num_groups, float thresh, int top_k, int32_t *keep,
                                 int32_t *keep_cnt, void *stream)
{
    if (!dets || !unit_group || !keep || !keep_cnt) return FCN_E_BADARG;
    if (num_units <= 0 || rows_per_unit <= 0 || num_groups <= 0 || top_k <= 0) return FCN_E_BADARG;
    NmsArgs a;
    a.dets = dets; a.valid = valid; a.unit_group = unit_group; a.U = num_units; a.rows_per_unit = rows_per_unit;
    a.G = num_groups; a.top_k = top_k; a.thresh = thresh; a.keep = keep; a.keep_cnt = keep_cnt;
    hipLaunchKernelGGL(nms_kernel, dim3(num_groups), dim3(NMS_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}
