// Rotated-box overlap core shared by the IoU training metrics (loss_tail.hip), the paired IoU entry point and the
// rotated 3-D NMS (box_iou.hip).
//
// Replaces the boost::geometry polygon clipping of ops/pybind11/box_ops.h:173-260 (rbbox_iou_3d_pair) and
// ops/pybind11/nms_cpu.h:148-240 (rotate_non_max_suppression_3d_cpu): both intersect the bird's-eye-view rectangles of two
// boxes (corners 6,7,4,5 of get_box3d_corners_helper, models/model_util.py:48-72, in the x-z plane) and multiply by the
// overlap of the y extents.  Two convex quadrilaterals need no general polygon library: Sutherland-Hodgman clipping of one
// by the four edges of the other is exact (at most 8 vertices), orientation-agnostic here, and runs in registers /
// private memory of one thread.  Same formulation as the reference's own pure-python utils/box_util.py:11-56,93-119,
// which is what pins it (tests/golden/make_golden_iou.py).
//
// Plain C++ float arithmetic only (no HIP intrinsics): tests/host_harness compiles this header with g++ to check the
// arithmetic against the golden vectors on a machine without a GPU.  That harness is test infrastructure; the product
// calls these functions from device code only.
#pragma once

#ifndef FCN_HD
#define FCN_HD __host__ __device__ __forceinline__
#endif

#define FCN_CLIP_MAXV 10

// Twice the signed area of a polygon (shoelace).
FCN_HD float fcn_poly_area2(const float *x, const float *z, int n)
{
    float s = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        s += x[i] * z[j] - x[j] * z[i];
    }
    return s;
}

// Area of the intersection of two convex quadrilaterals given as cyclic vertex lists (either orientation).
FCN_HD float fcn_quad_intersection_area(const float *ax, const float *az, const float *bx, const float *bz)
{
    float px[FCN_CLIP_MAXV], pz[FCN_CLIP_MAXV], qx[FCN_CLIP_MAXV], qz[FCN_CLIP_MAXV];
    int n = 4;
    for (int i = 0; i < 4; ++i) { px[i] = ax[i]; pz[i] = az[i]; }
    const float sgn = fcn_poly_area2(bx, bz, 4) >= 0.f ? 1.f : -1.f;      // interior side of b's edges
    for (int e = 0; e < 4; ++e) {
        const float c1x = bx[e], c1z = bz[e];
        const float ex = bx[(e + 1) & 3] - c1x, ez = bz[(e + 1) & 3] - c1z;
        int m = 0;
        float sx = px[n - 1], sz = pz[n - 1];
        float ds = sgn * (ex * (sz - c1z) - ez * (sx - c1x));
        for (int i = 0; i < n; ++i) {
            const float vx = px[i], vz = pz[i];
            const float dv = sgn * (ex * (vz - c1z) - ez * (vx - c1x));
            const bool in_v = dv > 0.f, in_s = ds > 0.f;                  // strict, as utils/box_util.py:26-27
            if (in_v != in_s && m < FCN_CLIP_MAXV) {                      // the edge s -> v crosses the clip line
                const float t = ds / (ds - dv);
                qx[m] = sx + t * (vx - sx);
                qz[m] = sz + t * (vz - sz);
                ++m;
            }
            if (in_v && m < FCN_CLIP_MAXV) { qx[m] = vx; qz[m] = vz; ++m; }
            sx = vx; sz = vz; ds = dv;
        }
        n = m;
        if (n == 0) return 0.f;
        for (int i = 0; i < n; ++i) { px[i] = qx[i]; pz[i] = qz[i]; }
    }
    if (n < 3) return 0.f;
    const float a2 = fcn_poly_area2(px, pz, n);
    return 0.5f * (a2 < 0.f ? -a2 : a2);
}

// Bird's-eye-view rectangle of a box in the vertex order of the reference's polygons (corners 6,7,4,5):
// (-l/2,-w/2), (-l/2,+w/2), (+l/2,+w/2), (+l/2,-w/2) rotated by the heading about y and moved to (cx, cz).
FCN_HD void fcn_bev_rect(float cx, float cz, float l, float w, float co, float si, float *x, float *z)
{
    const float hx[4] = {-0.5f * l, -0.5f * l, 0.5f * l, 0.5f * l};
    const float hz[4] = {-0.5f * w, 0.5f * w, 0.5f * w, -0.5f * w};
    for (int i = 0; i < 4; ++i) {
        x[i] = co * hx[i] + si * hz[i] + cx;
        z[i] = -si * hx[i] + co * hz[i] + cz;
    }
}

// (BEV IoU, 3-D IoU) of two boxes from their BEV polygons and y extents (ytop = corner 0's y, ybot = corner 4's y: y points
// down in camera coordinates, so ytop > ybot), with the reference's formulas (box_ops.h:230-249):
//   inter_vol = inter_area * max(0, min(ytop) - max(ybot)); vol = max(0, area * (ytop - ybot));
//   iou2d = inter_area / union_area (= area_a + area_b - inter_area), iou3d = inter_vol / (vol_a + vol_b - inter_vol).
FCN_HD void fcn_iou_from_polys(const float *ax, const float *az, float a_ytop, float a_ybot, const float *bx, const float *bz,
                               float b_ytop, float b_ybot, float *iou2d, float *iou3d)
{
    *iou2d = 0.f;
    *iou3d = 0.f;
    const float inter = fcn_quad_intersection_area(ax, az, bx, bz);
    if (!(inter > 0.f)) return;                                          // empty intersection: the reference leaves zeros
    float aa = 0.5f * fcn_poly_area2(ax, az, 4), ab = 0.5f * fcn_poly_area2(bx, bz, 4);
    aa = aa < 0.f ? -aa : aa;
    ab = ab < 0.f ? -ab : ab;
    const float uni = aa + ab - inter;
    const float ymax = a_ytop < b_ytop ? a_ytop : b_ytop, ymin = a_ybot > b_ybot ? a_ybot : b_ybot;
    const float dy = ymax - ymin;
    const float ivol = inter * (dy > 0.f ? dy : 0.f);
    float va = aa * (a_ytop - a_ybot), vb = ab * (b_ytop - b_ybot);
    va = va > 0.f ? va : 0.f;
    vb = vb > 0.f ? vb : 0.f;
    *iou2d = inter / uni;
    *iou3d = ivol / (va + vb - ivol);
}

// The same from box parameters (centre, size (l, w, h), cos / sin of the heading).
FCN_HD void fcn_iou_from_params(float acx, float acy, float acz, float al, float aw, float ah, float aco, float asi,
                                float bcx, float bcy, float bcz, float bl, float bw, float bh, float bco, float bsi,
                                float *iou2d, float *iou3d)
{
    float ax[4], az[4], bx[4], bz[4];
    fcn_bev_rect(acx, acz, al, aw, aco, asi, ax, az);
    fcn_bev_rect(bcx, bcz, bl, bw, bco, bsi, bx, bz);
    fcn_iou_from_polys(ax, az, acy + 0.5f * ah, acy - 0.5f * ah, bx, bz, bcy + 0.5f * bh, bcy - 0.5f * bh, iou2d, iou3d);
}
