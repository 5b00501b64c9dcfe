// Split-encoded weight images of a PointNet scale (device code shared by pointnet_fwd.hip -- standalone packing launches -- and
// group_compact.hip, whose first launch packs the weights of all scales on the side: no launch of its own on the step's chain).
#pragma once
#include "gemm_tile.h"

// ------------------------------------------------------------------------------------------------
// Encoded weight images of one scale (fcn_pn_ws.wenc; gemm_tile.h "kb-major").  Weights change once per step, the GEMMs read
// them hundreds of times: they are split-encoded ONCE per step into the register image of the MFMA operand, so the GEMM loops
// stage them with plain 16-byte copies (no VALU).  Four images, each as many floats as its weight:
//   F2, F3  forward operand of conv2 / conv3 (reduction over Cin):  [Cin/32][plane][4][Cout] u32x4 <- W[n][32c + 8kb + 0..7]
//   G2, G3  data-gradient operand (reduction over Cout):            [Cout/32][plane][4][Cin] u32x4 <- W[32c + 8kb + 0..7][k]
// Offsets in floats: F2 0, F3 C2*C1, G2 C2*C1 + C3*C2, G3 2*C2*C1 + C3*C2.
__host__ __device__ inline int64_t pn_wenc_off(int img, int C1, int C2, int C3)
{
    const int64_t a = (int64_t)C2 * C1, b = (int64_t)C3 * C2;
    return img == 0 ? 0 : (img == 1 ? a : (img == 2 ? a + b : 2 * a + b));
}
__host__ __device__ inline int64_t pn_wenc_floats(int C1, int C2, int C3) { return 2 * ((int64_t)C2 * C1 + (int64_t)C3 * C2); }

// item t of one image: one (chunk, k-block, column) -> both planes
template <int MM>
__device__ __forceinline__ void pn_pack_item(const float *__restrict__ W, int COUT, int CIN, bool grad, int t, u32x4 *__restrict__ img)
{
    const int NC = grad ? CIN : COUT;                   // columns of the image
    const int n = t % NC, r = t / NC, kb = r & 3, c = r >> 2;
    float x[8];
    if (!grad) {
        const v4f a = ldg4(W + (int64_t)n * CIN + c * KC + 8 * kb), b = ldg4(W + (int64_t)n * CIN + c * KC + 8 * kb + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = W[(int64_t)(c * KC + 8 * kb + j) * CIN + n];
    }
    u32x4 hi, lo;
    enc8<MM>(x, hi, lo);
    img[((int64_t)c * 8 + kb) * NC + n] = hi;
    img[((int64_t)c * 8 + 4 + kb) * NC + n] = lo;
}

struct PackArgs {
    const float *W2, *W3;
    float *wenc;
    int C1, C2, C3, precision;
};

// items of a scale: each image has Cout*Cin/8 of them
__device__ __forceinline__ void pn_pack_range(const PackArgs &a, int first, int step)
{
    const int n2 = a.C2 * a.C1 / 8, n3 = a.C3 * a.C2 / 8;
    u32x4 *F2 = (u32x4 *)(a.wenc + pn_wenc_off(0, a.C1, a.C2, a.C3)), *F3 = (u32x4 *)(a.wenc + pn_wenc_off(1, a.C1, a.C2, a.C3));
    u32x4 *G2 = (u32x4 *)(a.wenc + pn_wenc_off(2, a.C1, a.C2, a.C3)), *G3 = (u32x4 *)(a.wenc + pn_wenc_off(3, a.C1, a.C2, a.C3));
    const int mmf = FCN_MM_OF(a.precision, true), mmb = FCN_MM_OF(a.precision, false);
    for (int t = first; t < 2 * (n2 + n3); t += step) {
        const bool grad = t >= n2 + n3;
        const int u = grad ? t - (n2 + n3) : t;
        const bool l3 = u >= n2;
        const int v = l3 ? u - n2 : u;
        const float *W = l3 ? a.W3 : a.W2;
        const int COUT = l3 ? a.C3 : a.C2, CIN = l3 ? a.C2 : a.C1;
        u32x4 *img = grad ? (l3 ? G3 : G2) : (l3 ? F3 : F2);
        const int mm = grad ? mmb : mmf;
        if (mm == MM_F32) pn_pack_item<MM_F32>(W, COUT, CIN, grad, v, img);
        else if (mm == MM_F16X3) pn_pack_item<MM_F16X3>(W, COUT, CIN, grad, v, img);
        else if (mm == MM_BF16X3) pn_pack_item<MM_BF16X3>(W, COUT, CIN, grad, v, img);
        else pn_pack_item<MM_BF16X1>(W, COUT, CIN, grad, v, img);
    }
}

