// On-device construction of one training batch from raw frustum records: what the reference's data loader does per
// sample on the host in numpy (datasets/provider_sample.py::ProviderDataset.__getitem__ :137-262 with generate_ref
// :291-327, generate_labels :270-289, centre-view helpers :329-372; datasets/data_utils.py rotate_pc_along_y :7-21,
// compute_box_3d :44-70, project_image_to_rect :73-93) and collates into the dict PointNetDet.forward consumes.
// One workgroup per frustum; HBM/latency-bound gather + a few hundred fp64 operations -- no GEMM in sight.
//   points:  resample (indices drawn on the host: the reference's np.random.choice), rotate to the frustum's centre
//            view, optional x-flip and depth shift, written channel-major (B,3,N) with lanes along N
//   centres: arange(0, max_depth, stride) + stride/2 on the ray through the 2-D box centre (calibration P), rotated
//   labels:  +1 inside the half-size box, -1 inside the full box, nearest centre when none is inside the half box
// fp64 where numpy computes in fp64, rounded to fp32 exactly where the reference stores fp32.
// The same kernel serves the SUN-RGBD loader (datasets/provider_sample_sunrgbd.py::ProviderDataset.__getitem__ :116-263 with
// generate_ref :283-326 and project_image_to_upright_camera :28-59): five strides, window centres through the camera
// matrix K and the tilt rotation Rtilt instead of the KITTI projection P, and an extra height shift in the augmentation.
#include "fcn_common.h"
#include "../../include/fcn_hip.h"

#define INP_T 256

struct InpDescG {              // both entry points in one shape
    int B, N, pt_stride, nsc;
    int L[5];
    double stride[5], max_depth;
    int random_flip, random_shift;
};

struct InpArgs {
    InpDescG d;
    const float *raw;
    const int64_t *off, *raw_seg;
    const int32_t *choice;
    const double *fangle, *box2d, *P, *corners, *heading, *size, *coin, *normal;
    const double *K, *Rtilt;   // SUN-RGBD: (B,3,3) camera matrix and tilt rotation (P == nullptr then)
    const double *hshift;      // SUN-RGBD: the uniform [0,1) draw of the height shift (random_shift), else nullptr
    float *pc, *ref[5];
    int64_t *cls;
    float *center, *head, *osize, *rot;
    int64_t *seg;
};

__device__ __forceinline__ bool inp_in_box(double dx, double dy, double dz, double c, double s, double l, double w, double h)
{
    const double lx = c * dx - s * dz, lz = s * dx + c * dz;
    return fabs(lx) <= l / 2.0 && fabs(dy) <= h / 2.0 && fabs(lz) <= w / 2.0;
}

__global__ __launch_bounds__(INP_T) void prepare_inputs_kernel(InpArgs a)
{
    __shared__ double sdist[INP_T];
    __shared__ int sidx[INP_T];
    __shared__ int sany;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = a.d.N;
    // ---- per-frustum scalars (every thread computes the same values)
    const double rot = M_PI / 2.0 + a.fangle[b];
    const double c = cos(rot), s = sin(rot);
    const double *cr = a.corners + (int64_t)b * 24;
    const double c0x = (cr[0] + cr[18]) / 2.0, c0y = (cr[1] + cr[19]) / 2.0, c0z = (cr[2] + cr[20]) / 2.0;
    double cx = c0x * c + c0z * (-s), cy = c0y, cz = c0x * s + c0z * c;     // (cy: the height shift below moves it)
    double ang = a.heading[b] - rot;
    const bool flip = a.d.random_flip && a.coin[b] > 0.5;
    if (flip) { cx = -cx; ang = M_PI - ang; }
    const double sl = a.size[3 * b], sw = a.size[3 * b + 1], sh = a.size[3 * b + 2];
    double shift = 0.0;
    if (a.d.random_shift) {
        const double dist = sqrt(sl * sl + sw * sw);
        shift = fmin(fmax(a.normal[b] * dist * 0.2, -0.5 * dist), 0.5 * dist);
        shift = fmin(fmax(shift + cz, 0.0), a.d.max_depth) - cz;
        cz += shift;
    }
    // provider_sample_sunrgbd.py:228-230: height_shift = np.random.random() * 0.4 - 0.2 on the points' y and the box centre
    const bool has_h = a.d.random_shift && a.hshift != nullptr;
    const double hsh = has_h ? a.hshift[b] * 0.4 - 0.2 : 0.0;
    if (has_h) cy += hsh;
    if (tid == 0) {
        a.center[3 * b] = (float)cx; a.center[3 * b + 1] = (float)cy; a.center[3 * b + 2] = (float)cz;
        a.head[b] = (float)ang;
        a.osize[3 * b] = (float)sl; a.osize[3 * b + 1] = (float)sw; a.osize[3 * b + 2] = (float)sh;
        a.rot[b] = (float)rot;
    }
    // ---- points
    const int64_t o0 = a.off[b];
    const int ps = a.d.pt_stride;
    for (int i = tid; i < N; i += INP_T) {
        const int64_t j = o0 + a.choice[(int64_t)b * N + i];
        const float *p = a.raw + j * ps;
        const double x = p[0], z = p[2];
        float xr = (float)(x * c + z * (-s));              // the reference stores the rotated record back as float32
        float zr = (float)(x * s + z * c);
        if (flip) xr = -xr;
        if (a.d.random_shift) zr = (float)((double)zr + shift);
        float yr = p[1];
        if (has_h) yr = (float)((double)yr + hsh);         // float32 record += float64 scalar
        float *o = a.pc + (int64_t)b * 3 * N;
        o[i] = xr; o[N + i] = yr; o[2 * N + i] = zr;
        if (a.seg) a.seg[(int64_t)b * N + i] = a.raw_seg[j];
    }
    // ---- frustum centres of the strides, labels on stride 2
    const bool upright = a.P == nullptr;
    double cu, cv, fu, fv, bx = 0.0, by = 0.0;
    double Rt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!upright) {
        const double *P = a.P + (int64_t)b * 12;
        cu = P[2]; cv = P[6]; fu = P[0]; fv = P[5];
        bx = P[3] / (-fu); by = P[7] / (-fv);
    } else {
        const double *K = a.K + (int64_t)b * 9;
        cu = K[2]; cv = K[5]; fu = K[0]; fv = K[4];
        for (int i = 0; i < 9; ++i) Rt[i] = a.Rtilt[(int64_t)b * 9 + i];
    }
    const double u0 = (a.box2d[4 * b] + a.box2d[4 * b + 2]) / 2.0, v0 = (a.box2d[4 * b + 1] + a.box2d[4 * b + 3]) / 2.0;
    const double ca = cos(ang), sa = sin(ang);
    double best = 1e300;
    int bidx = 0x7fffffff;
    bool any1 = false;
    if (tid == 0) sany = 0;
    __syncthreads();
#pragma unroll 1
    for (int sc = 0; sc < a.d.nsc; ++sc) {
        const int L = a.d.L[sc];
        const double st = a.d.stride[sc];
        for (int l = tid; l < L; l += INP_T) {
            const double zd = (double)l * st + st / 2.0;
            double x = ((u0 - cu) * zd) / fu + bx, y = ((v0 - cv) * zd) / fv + by, z = zd;
            if (upright) {
                // project_image_to_upright_camera (provider_sample_sunrgbd.py:44-59): camera (x, y, z) -> depth frame
                // (x, z, -y) -> Rtilt . -> (X, -Z, Y)
                const double d0 = x, d1 = zd, d2 = -y;
                const double u_0 = Rt[0] * d0 + Rt[1] * d1 + Rt[2] * d2;
                const double u_1 = Rt[3] * d0 + Rt[4] * d1 + Rt[5] * d2;
                const double u_2 = Rt[6] * d0 + Rt[7] * d1 + Rt[8] * d2;
                x = u_0; y = -u_2; z = u_1;
            }
            double xr = x * c + z * (-s);
            const double zr = x * s + z * c;
            if (flip) xr = -xr;
            float *o = a.ref[sc] + (int64_t)b * 3 * L;
            o[l] = (float)xr; o[L + l] = (float)y; o[2 * L + l] = (float)zr;
            if (sc == 1 && a.cls) {
                const double dx = xr - cx, dy = y - cy, dz = zr - cz;
                const bool in1 = inp_in_box(dx, dy, dz, ca, sa, sl * 0.5, sw * 0.5, sh * 0.5);
                const bool in2 = inp_in_box(dx, dy, dz, ca, sa, sl, sw, sh);
                a.cls[(int64_t)b * L + l] = in1 ? 1 : (in2 ? -1 : 0);
                any1 = any1 || in1;
                const double dd = sqrt(dx * dx + dy * dy + dz * dz);
                if (dd < best) { best = dd; bidx = l; }       // l ascends per thread: first minimum kept
            }
        }
    }
    if (!a.cls) return;
    if (any1) sany = 1;
    sdist[tid] = best; sidx[tid] = bidx;
    __syncthreads();
    if (sany) return;
    // nobody inside the half-size box: the nearest centre is the positive (np.argmin: first of equal minima)
    for (int o = INP_T / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const double d2 = sdist[tid + o];
            const int i2 = sidx[tid + o];
            if (d2 < sdist[tid] || (d2 == sdist[tid] && i2 < sidx[tid])) { sdist[tid] = d2; sidx[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0 && sidx[0] != 0x7fffffff) a.cls[(int64_t)b * a.d.L[1] + sidx[0]] = 1;
}

extern "C" int fcn_prepare_inputs(const fcn_inp_desc *d, const float *raw_pts, const int64_t *pt_off, const int64_t *raw_seg,
                                  const int32_t *choice, const double *frustum_angle, const double *box2d, const double *P,
                                  const double *box3d_corners, const double *heading, const double *size,
                                  const double *coin, const double *normal, float *point_cloud, float *const center_ref[4],
                                  int64_t *cls_label, float *box3d_center, float *box3d_heading, float *box3d_size,
                                  float *rot_angle, int64_t *seg_label, void *stream)
{
    if (!d || !raw_pts || !pt_off || !choice || !frustum_angle || !box2d || !P || !point_cloud || !center_ref ||
        !box3d_center || !box3d_heading || !box3d_size || !rot_angle)
        return FCN_E_BADARG;
    if (!box3d_corners || !heading || !size) return FCN_E_BADARG;
    if (d->B <= 0 || d->N <= 0 || d->pt_stride < 3) return FCN_E_BADARG;
    if ((d->random_flip && !coin) || (d->random_shift && !normal) || (seg_label && !raw_seg)) return FCN_E_BADARG;
    for (int s = 0; s < 4; ++s)
        if (d->L[s] <= 0 || !(d->stride[s] > 0.0) || !center_ref[s]) return FCN_E_BADARG;
    InpArgs a;
    a.d.B = d->B; a.d.N = d->N; a.d.pt_stride = d->pt_stride; a.d.nsc = 4; a.d.max_depth = d->max_depth;
    a.d.random_flip = d->random_flip; a.d.random_shift = d->random_shift;
    for (int s = 0; s < 5; ++s) { a.d.L[s] = s < 4 ? d->L[s] : 0; a.d.stride[s] = s < 4 ? d->stride[s] : 1.0; a.ref[s] = s < 4 ? center_ref[s] : nullptr; }
    a.raw = raw_pts; a.off = pt_off; a.raw_seg = raw_seg; a.choice = choice; a.fangle = frustum_angle;
    a.box2d = box2d; a.P = P; a.corners = box3d_corners; a.heading = heading; a.size = size; a.coin = coin; a.normal = normal;
    a.K = nullptr; a.Rtilt = nullptr; a.hshift = nullptr;
    a.pc = point_cloud;
    a.cls = cls_label; a.center = box3d_center; a.head = box3d_heading; a.osize = box3d_size; a.rot = rot_angle; a.seg = seg_label;
    hipLaunchKernelGGL(prepare_inputs_kernel, dim3(d->B), dim3(INP_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// SUN-RGBD loader (cfgs/det_sample_sunrgbd.yaml; datasets/provider_sample_sunrgbd.py:116-326): five strides, centres through
// K and Rtilt, depth + height shift.
extern "C" int fcn_prepare_inputs_sunrgbd(const fcn_inp5_desc *d, const float *raw_pts, const int64_t *pt_off,
                                          const int64_t *raw_seg, const int32_t *choice, const double *frustum_angle,
                                          const double *box2d, const double *K, const double *Rtilt,
                                          const double *box3d_corners, const double *heading, const double *size,
                                          const double *coin, const double *normal, const double *hshift,
                                          float *point_cloud, float *const center_ref[5], int64_t *cls_label,
                                          float *box3d_center, float *box3d_heading, float *box3d_size, float *rot_angle,
                                          int64_t *seg_label, void *stream)
{
    if (!d || !raw_pts || !pt_off || !choice || !frustum_angle || !box2d || !K || !Rtilt || !point_cloud || !center_ref ||
        !box3d_center || !box3d_heading || !box3d_size || !rot_angle)
        return FCN_E_BADARG;
    if (!box3d_corners || !heading || !size) return FCN_E_BADARG;
    if (d->B <= 0 || d->N <= 0 || d->pt_stride < 3) return FCN_E_BADARG;
    if ((d->random_flip && !coin) || (d->random_shift && (!normal || !hshift)) || (seg_label && !raw_seg)) return FCN_E_BADARG;
    for (int s = 0; s < 5; ++s)
        if (d->L[s] <= 0 || !(d->stride[s] > 0.0) || !center_ref[s]) return FCN_E_BADARG;
    InpArgs a;
    a.d.B = d->B; a.d.N = d->N; a.d.pt_stride = d->pt_stride; a.d.nsc = 5; a.d.max_depth = d->max_depth;
    a.d.random_flip = d->random_flip; a.d.random_shift = d->random_shift;
    for (int s = 0; s < 5; ++s) { a.d.L[s] = d->L[s]; a.d.stride[s] = d->stride[s]; a.ref[s] = center_ref[s]; }
    a.raw = raw_pts; a.off = pt_off; a.raw_seg = raw_seg; a.choice = choice; a.fangle = frustum_angle;
    a.box2d = box2d; a.P = nullptr; a.corners = box3d_corners; a.heading = heading; a.size = size; a.coin = coin; a.normal = normal;
    a.K = K; a.Rtilt = Rtilt; a.hshift = d->random_shift ? hshift : nullptr;
    a.pc = point_cloud;
    a.cls = cls_label; a.center = box3d_center; a.head = box3d_heading; a.osize = box3d_size; a.rot = rot_angle; a.seg = seg_label;
    hipLaunchKernelGGL(prepare_inputs_kernel, dim3(d->B), dim3(INP_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Refinement stage (cfgs/refine_car.yaml; datasets/provider_sample_refine.py::ProviderDataset.__getitem__ :176-315 with
// get_center_view_point / _box3d :137-152, generate_ref :336-386, generate_labels :317-334, and collate_fn :388-419).  The
// sample is normalised to the FIRST-STAGE PREDICTION: points and label box are translated to the predicted box centre and
// rotated by its heading; the window centres span the predicted box's own depth extent -- in that frame the box is
// axis-aligned at the origin, so they are (0, 0, -w/2 + l*s + s/2) for l < ceil(w / s) (np.arange(z1, z2, s)) -- a different
// count per sample, which the reference's collate_fn pads by repeating the last centre / label up to the batch maximum
// (Lpad, handed in by the caller).  Positive / ignore boxes are the label box scaled by 0.3 / 0.6.
struct InpRefineArgs {
    fcn_inp_refine_desc d;
    const float *raw;
    const int64_t *off;
    const int32_t *choice;
    const double *pred_corners, *pred_angle, *pred_size, *corners, *heading, *size, *coin, *normal;
    float *pc, *ref[4];
    int64_t *cls;
    float *center, *head, *osize, *rot, *refc;
    int32_t *lens;
};

__global__ __launch_bounds__(INP_T) void prepare_inputs_refine_kernel(InpRefineArgs a)
{
    __shared__ double sdist[INP_T];
    __shared__ int sidx[INP_T];
    __shared__ int sany;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = a.d.N;
    const double *pcn = a.pred_corners + (int64_t)b * 24;
    const double pcx = (pcn[0] + pcn[18]) / 2.0, pcy = (pcn[1] + pcn[19]) / 2.0, pcz = (pcn[2] + pcn[20]) / 2.0;
    const double rot = a.pred_angle[b];
    const double c = cos(rot), s = sin(rot);
    const bool train = a.corners != nullptr;
    const bool flip = train && a.d.random_flip && a.coin[b] > 0.5;
    double cx = 0.0, cy = 0.0, cz = 0.0, ang = 0.0, sl = 0.0, sw = 0.0, sh = 0.0, shift = 0.0;
    if (train) {
        const double *cr = a.corners + (int64_t)b * 24;
        const double dx = (cr[0] + cr[18]) / 2.0 - pcx, dy = (cr[1] + cr[19]) / 2.0 - pcy, dz = (cr[2] + cr[20]) / 2.0 - pcz;
        cx = dx * c + dz * (-s); cy = dy; cz = dx * s + dz * c;
        ang = a.heading[b] - rot;
        if (flip) { cx = -cx; ang = M_PI - ang; }
        sl = a.size[3 * b]; sw = a.size[3 * b + 1]; sh = a.size[3 * b + 2];
        if (a.d.random_shift) {
            const double dist = sqrt(sl * sl + sw * sw), s1 = a.d.stride[0];
            shift = fmin(fmax(a.normal[b] * dist * 0.1, -2.0 * s1), 2.0 * s1);
            cz += shift;
        }
    }
    if (tid == 0) {
        if (train) {
            a.center[3 * b] = (float)cx; a.center[3 * b + 1] = (float)cy; a.center[3 * b + 2] = (float)cz;
            a.head[b] = (float)ang;
            a.osize[3 * b] = (float)sl; a.osize[3 * b + 1] = (float)sw; a.osize[3 * b + 2] = (float)sh;
        }
        a.rot[b] = (float)rot;
        a.refc[3 * b] = (float)pcx; a.refc[3 * b + 1] = (float)pcy; a.refc[3 * b + 2] = (float)pcz;
    }
    // ---- points: (p - centre) rotated by the predicted heading, stored as float32
    const int64_t o0 = a.off[b];
    const int ps = a.d.pt_stride;
    for (int i = tid; i < N; i += INP_T) {
        const int64_t j = o0 + a.choice[(int64_t)b * N + i];
        const float *p = a.raw + j * ps;
        const double x = (double)p[0] - pcx, y = (double)p[1] - pcy, z = (double)p[2] - pcz;
        float xr = (float)(x * c + z * (-s));
        float zr = (float)(x * s + z * c);
        if (flip) xr = -xr;
        if (train && a.d.random_shift) zr = (float)((double)zr + shift);
        float *o = a.pc + (int64_t)b * 3 * N;
        o[i] = xr; o[N + i] = (float)y; o[2 * N + i] = zr;
    }
    // ---- window centres over the predicted box's depth extent, edge-padded to Lpad; labels on stride 2
    const double w = a.pred_size[3 * b + 1];
    const double z1 = -w / 2.0, z2 = w / 2.0;
    const double ca = cos(ang), sa = sin(ang);
    double best = 1e300;
    int bidx = 0x7fffffff;
    bool any1 = false;
    if (tid == 0) sany = 0;
    __syncthreads();
#pragma unroll 1
    for (int sc = 0; sc < 4; ++sc) {
        const int Lp = a.d.Lpad[sc];
        const double st = a.d.stride[sc];
        int Lb = (int)ceil((z2 - z1) / st);                         // len(np.arange(z1, z2, st))
        Lb = Lb < 0 ? 0 : (Lb > Lp ? Lp : Lb);
        if (tid == 0) a.lens[4 * b + sc] = Lb;
        for (int l = tid; l < Lp; l += INP_T) {
            const int le = l < Lb ? l : Lb - 1;                     // collate_fn: np.pad(..., mode='edge')
            const double z = (z1 + (double)le * st) + st / 2.0;
            const double xr = flip ? -0.0 : 0.0;
            float *o = a.ref[sc] + (int64_t)b * 3 * Lp;
            o[l] = (float)xr; o[Lp + l] = 0.f; o[2 * Lp + l] = (float)z;
            if (sc == 1 && a.cls && l < Lb) {
                const double dx = xr - cx, dy = 0.0 - cy, dz = z - cz;
                const bool in1 = inp_in_box(dx, dy, dz, ca, sa, sl * 0.3, sw * 0.3, sh * 0.3);
                const bool in2 = inp_in_box(dx, dy, dz, ca, sa, sl * 0.6, sw * 0.6, sh * 0.6);
                a.cls[(int64_t)b * Lp + l] = in1 ? 1 : (in2 ? -1 : 0);
                any1 = any1 || in1;
                const double dd = sqrt(dx * dx + dy * dy + dz * dz);
                if (dd < best) { best = dd; bidx = l; }
            }
        }
    }
    if (!a.cls) return;
    if (any1) sany = 1;
    sdist[tid] = best; sidx[tid] = bidx;
    __syncthreads();
    if (!sany) {
        for (int o = INP_T / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const double d2 = sdist[tid + o];
                const int i2 = sidx[tid + o];
                if (d2 < sdist[tid] || (d2 == sdist[tid] && i2 < sidx[tid])) { sdist[tid] = d2; sidx[tid] = i2; }
            }
            __syncthreads();
        }
        if (tid == 0 && sidx[0] != 0x7fffffff) a.cls[(int64_t)b * a.d.Lpad[1] + sidx[0]] = 1;
    }
    __syncthreads();
    __threadfence_block();
    // edge padding of the labels: positions >= L_b repeat the last real one
    {
        const int Lp = a.d.Lpad[1];
        int Lb = (int)ceil((z2 - z1) / a.d.stride[1]);
        Lb = Lb < 0 ? 0 : (Lb > Lp ? Lp : Lb);
        if (Lb > 0) {
            const int64_t last = a.cls[(int64_t)b * Lp + Lb - 1];
            for (int l = Lb + tid; l < Lp; l += INP_T) a.cls[(int64_t)b * Lp + l] = last;
        }
    }
}

extern "C" int fcn_prepare_inputs_refine(const fcn_inp_refine_desc *d, const float *raw_pts, const int64_t *pt_off,
                                         const int32_t *choice, const double *pred_corners, const double *pred_angle,
                                         const double *pred_size, const double *box3d_corners, const double *heading,
                                         const double *size, const double *coin, const double *normal, float *point_cloud,
                                         float *const center_ref[4], int64_t *cls_label, float *box3d_center,
                                         float *box3d_heading, float *box3d_size, float *rot_angle, float *ref_center,
                                         int32_t *lens, void *stream)
{
    if (!d || !raw_pts || !pt_off || !choice || !pred_corners || !pred_angle || !pred_size || !point_cloud || !center_ref ||
        !rot_angle || !ref_center || !lens)
        return FCN_E_BADARG;
    if (d->B <= 0 || d->N <= 0 || d->pt_stride < 3) return FCN_E_BADARG;
    const bool train = box3d_corners != nullptr;
    if (train && (!heading || !size || !box3d_center || !box3d_heading || !box3d_size)) return FCN_E_BADARG;
    if (!train && cls_label) return FCN_E_BADARG;
    if (train && ((d->random_flip && !coin) || (d->random_shift && !normal))) return FCN_E_BADARG;
    for (int s = 0; s < 4; ++s)
        if (d->Lpad[s] <= 0 || !(d->stride[s] > 0.0) || !center_ref[s]) return FCN_E_BADARG;
    InpRefineArgs a;
    a.d = *d; a.raw = raw_pts; a.off = pt_off; a.choice = choice; a.pred_corners = pred_corners; a.pred_angle = pred_angle;
    a.pred_size = pred_size; a.corners = box3d_corners; a.heading = heading; a.size = size; a.coin = coin; a.normal = normal;
    a.pc = point_cloud;
    for (int s = 0; s < 4; ++s) a.ref[s] = center_ref[s];
    a.cls = cls_label; a.center = box3d_center; a.head = box3d_heading; a.osize = box3d_size; a.rot = rot_angle;
    a.refc = ref_center; a.lens = lens;
    hipLaunchKernelGGL(prepare_inputs_refine_kernel, dim3(d->B), dim3(INP_T), 0, (hipStream_t)stream, a);
    FCN_CHECK_LAUNCH();
    return 0;
}
