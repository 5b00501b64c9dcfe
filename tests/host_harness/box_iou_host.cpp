// TEST HARNESS (not part of the product): compiles frustum_convnet_amd/csrc/box_iou.h -- the clip / IoU core the HIP
// kernels call from device code -- with g++ so that its float arithmetic can be checked against the golden vectors in the
// build container, which has no GPU.  Built on demand by tests/test_oracle_box.py into tests/host_harness/_build/.
#include <math.h>
#define FCN_HD static inline
#include "../../frustum_convnet_amd/csrc/box_iou.h"

extern "C" void host_iou_from_params(const float *a7, const float *b7, int n, float *out2)
{
    for (int i = 0; i < n; ++i) {
        const float *a = a7 + 7 * i, *b = b7 + 7 * i;
        fcn_iou_from_params(a[0], a[1], a[2], a[3], a[4], a[5], cosf(a[6]), sinf(a[6]),
                            b[0], b[1], b[2], b[3], b[4], b[5], cosf(b[6]), sinf(b[6]), out2 + 2 * i, out2 + 2 * i + 1);
    }
}

extern "C" void host_iou_from_corners(const float *c1, const float *c2, int n, float *out2)
{
    const int order[4] = {6, 7, 4, 5};
    for (int i = 0; i < n; ++i) {
        float ax[4], az[4], bx[4], bz[4];
        for (int k = 0; k < 4; ++k) {
            ax[k] = c1[(i * 8 + order[k]) * 3 + 0]; az[k] = c1[(i * 8 + order[k]) * 3 + 2];
            bx[k] = c2[(i * 8 + order[k]) * 3 + 0]; bz[k] = c2[(i * 8 + order[k]) * 3 + 2];
        }
        fcn_iou_from_polys(ax, az, c1[(i * 8 + 0) * 3 + 1], c1[(i * 8 + 4) * 3 + 1], bx, bz, c2[(i * 8 + 0) * 3 + 1],
                           c2[(i * 8 + 4) * 3 + 1], out2 + 2 * i, out2 + 2 * i + 1);
    }
}
