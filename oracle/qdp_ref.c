/*
 * ORACLE -- test infrastructure only.  Never imported, linked or executed by the
 * product path (frustum_convnet_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call into this file.
 *
 * CPU restatement of the reference's sliding-frustum grouping, following
 *   ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-65  (per-query serial scan)
 *   ops/query_depth_point/query_depth_point.py:29-40              (layouts, zero-init outputs)
 *
 * Semantics restated (kernel lines in brackets):
 *   - one query = one (sample b, centre m); only the z coordinate is read  [.cu:40,48]
 *   - scan points k = 0..n-1 in ascending index order                      [.cu:42]
 *   - stop as soon as nsample hits have been taken                         [.cu:44-45]
 *   - hit iff fabsf(z_centre - z_point) < dis_z, all in fp32, strict <     [.cu:51-53]
 *   - the first hit is written to every slot, hit #c then overwrites slot c [.cu:55-60]
 *   - pts_cnt = number of hits taken; untouched outputs stay zero          [.cu:64, .py:36-37]
 *   - idx is int64 (b,m,nsample), pts_cnt int32 (b,m)                      [.cu:19, .py:36-37]
 *
 * Pinning: the reference holds no CPU implementation and no asserted test for this op
 * (its CUDA source needs nvcc + THC headers absent from this image: unbuildable here).
 * The only reference artefact is ops/query_depth_point/test.py:10-28, whose mask
 * criterion abs(z - z1) < 0.2 is reproduced in tests/test_oracle_grouping.py.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -o libqdp_ref.so qdp_ref.c   (see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* pts: (b,n) z values at stride pt_stride floats, batch stride pt_bstride floats.
 * ctr: (b,m) z values likewise.  Strides let the caller pass either the (B,3,N)
 * layout (z row contiguous) or the kernel's (B,N,3) layout without copying. */
int qdp_ref_f32(const float *pts_z, int64_t pt_stride, int64_t pt_bstride,
                const float *ctr_z, int64_t ct_stride, int64_t ct_bstride,
                int b, int n, int m, float dis_z, int nsample,
                int64_t *idx, int32_t *cnt)
{
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return 1;
    memset(idx, 0, sizeof(int64_t) * (size_t)b * (size_t)m * (size_t)nsample);
    memset(cnt, 0, sizeof(int32_t) * (size_t)b * (size_t)m);
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int mi = 0; mi < m; ++mi) {
            const float *pz = pts_z + (int64_t)bi * pt_bstride;
            float z2 = ctr_z[(int64_t)bi * ct_bstride + (int64_t)mi * ct_stride];
            int64_t *row = idx + ((int64_t)bi * m + mi) * nsample;
            int c = 0;
            for (int k = 0; k < n; ++k) {
                if (c == nsample) break;
                float z1 = pz[(int64_t)k * pt_stride];
                float d3 = fabsf(z2 - z1);
                if (d3 < dis_z) {
                    if (c == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[c] = k;
                    c += 1;
                }
            }
            cnt[(int64_t)bi * m + mi] = c;
        }
    }
    return 0;
}
