// Fused train-loss tail of PointNetDet.forward (models/det_base.py:373-476 with models/common.py:217-232,
// models/model_util.py:9-19,48-72, models/box_transform.py:5-65): softmax focal loss over the non-ignored
// rows, and on the foreground rows (cls_label == 1) centre / heading / size / corner losses -- forward values
// AND d(total_loss)/d(logits) in one launch.  The reference runs ~150 tiny elementwise kernels and three host
// synchronisations here; on an MI355X that tail cost more than the whole fused PointNet forward.
//
// One row per thread, 128-thread workgroups (B*L2 ~ 4.5 k rows, of which ~B are foreground).  Every workgroup first
// counts the foreground rows itself (each mean is 1/nfg; 36 KB of labels from L2, cheaper than a second launch),
// then evaluates its rows, block-reduces the 8 loss sums + 3 accuracy counters and adds them to an accumulator with
// device-scope atomics; the last workgroup to arrive (ticket) turns the sums into the final scalars.
#include "fcn_common.h"

#define LT_NB 12          // heading bins (cfg.DATA.NUM_HEADING_BIN default, det_base.py:245)
// The kernel is instantiated for NS = 3 size clusters (KITTI, 41 logits per row, 128-thread workgroups) and NS = 10
// (SUN-RGBD, models/det_base_sunrgbd.py:271: 69 logits per row in 128-column rows, 64-thread workgroups -- the two staged
// rows per thread must fit the LDS)
#define LT_MIN_THREADS 64

struct LossArgs {
    const float *cls_raw;      // (B,2,L2)
    const float *reg_raw;      // (B,3+2*NB+4*NS,L2)
    const int64_t *cls_label;  // (B,L2)  in {-1,0,1}
    const float *ref2;         // (B,3,L2)
    const float *box_center;   // (B,3)
    const float *box_heading;  // (B,1)
    const float *box_size;     // (B,3)
    const int64_t *size_class; // (B,1)
    const float *mean_size;    // (NS,3)
    float *out;                // 16: total, cls, center, head_cls, head_res, size_cls, size_res, corners, cls_acc, head_acc, size_acc, nfg
    float *dcls;               // (B,2,L2)  d total / d cls_raw
    float *dreg;               // (B,39,L2) d total / d reg_raw
    int B, L2;
    float w_box, w_corner, w_headreg, w_sizereg;
    int ld;                    // 0: (B,C,L2) planes above; > 0: row-major logits (B*L2, ld) in cls_raw (cols 0..1 cls,
                               // 2.. reg) and row-major gradient in dcls (all ld columns written)
    // Optional persistent scratch (zeroed ONCE by the caller): [0] arrival ticket (int, reset by the last workgroup),
    // [32 + 16*g ..] the partial sums of workgroup g.  With it the launch needs no memset in front, and the final sums are
    // taken in workgroup order -- the reported scalars are reproducible bit for bit.
    float *scratch;
    float *total;              // optional copy of out[0] in its own buffer (the differentiable scalar of the binding)
};

// N sums at once through ONE barrier pair: v[i] <- the sum over the workgroup, as block_sum computes it (wave sum, then the waves
// in order).  sh: NT / 64 * N floats.
template <int NT, int N>
__device__ __forceinline__ void block_sums(float (&v)[N], float *sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum_f32(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) sh[wave * N + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += sh[w * N + i];
        v[i] = t;
    }
}

__device__ __forceinline__ float pymod(float a, float b) { return a - b * floorf(a / b); }

__device__ __forceinline__ void corners8(float cx, float cy, float cz, float c, float s, float l, float w, float h,
                                         float (&px)[8], float (&py)[8], float (&pz)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float sx = (k & 2) ? -0.5f : 0.5f;                         // + + - - + + - -
        const float sy = (k & 4) ? -0.5f : 0.5f;                         // + + + + - - - -
        const float sz = ((k & 3) == 1 || (k & 3) == 2) ? -0.5f : 0.5f;  // + - - + + - - +
        const float x = sx * l, y = sy * h, z = sz * w;
        px[k] = c * x + s * z + cx;
        py[k] = y + cy;
        pz[k] = -s * x + c * z + cz;
    }
}

template <int NS, int LT_THREADS>
__global__ __launch_bounds__(LT_THREADS) void loss_tail_kernel(LossArgs a)
{
    constexpr int NB = LT_NB, NC = 3 + 2 * NB + 4 * NS;
    constexpr int SW = (NC + 2) | 1;       // LDS row stride: the row's 2 + NC logits, odd
    __shared__ float shN[LT_THREADS / 64 * 11];
    __shared__ int last_s;
    // row-major variant: the workgroup's LT_THREADS gradient rows are staged here (odd stride: a thread writes its own
    // row without bank conflicts) and go out as coalesced 256-byte rows instead of 64 strided dwords per thread
    __shared__ float gS[LT_THREADS * SW];
    // the row's regression logits and its gradient row are indexed with run-time bins (heading class, size cluster): as
    // per-thread arrays they end up in scratch (320 B per lane, every access a trip to memory -- the kernel spent most of its
    // 50 us there); as LDS rows (odd stride, conflict-free) dynamic indexing is free
    __shared__ float gG[LT_THREADS * SW];
    const int tid = threadIdx.x;
    const int B = a.B, L2 = a.L2, R = B * L2;
    const float TWO_PI = 6.283185307179586f, PI = 3.141592653589793f;
    const float per = (float)(6.283185307179586 / NB), half = (float)(6.283185307179586 / NB / 2.0);

    // The workgroup's logits rows (row-major variant) are REQUESTED FIRST, as 16-byte loads into registers, so that their
    // trip overlaps the label count below; they go to LDS after it.  (One dword per loop iteration, as before, was a chain of
    // ~64 dependent memory round trips per thread: most of the kernel's 46 us on the critical path between the FCN's forward
    // and backward.)
    typedef float lt_v4f __attribute__((ext_vector_type(4)));
    typedef const lt_v4f __attribute__((address_space(1))) *lt_gv4fp;
    constexpr int LT_NV = 16;                            // float4 per thread per batch
    const int row0 = blockIdx.x * LT_THREADS;
    const int nrow = min(LT_THREADS, R - row0);
    const int q4 = a.ld >> 2;                            // float4 per row: 16 or 32
    const int sh_q = a.ld == 128 ? 5 : 4;
    const int nbatch = a.ld ? (q4 + LT_NV - 1) / LT_NV : 0;       // 1 (ld 64) or 2 (ld 128)
    lt_v4f lv[LT_NV];
#define LT_LOAD_BATCH(bt)                                                                                             \
    _Pragma("unroll") for (int k = 0; k < LT_NV; ++k) {                                                               \
        const int i = tid + LT_THREADS * (k + LT_NV * (bt));                                                          \
        const int rr = min(i >> sh_q, nrow - 1), c4 = i & (q4 - 1);     /* clamped row: unconditional loads */         \
        lv[k] = *(lt_gv4fp)(a.cls_raw + (int64_t)(row0 + rr) * a.ld + 4 * c4);                                        \
    }
#define LT_STAGE_BATCH(bt)                                                                                            \
    _Pragma("unroll") for (int k = 0; k < LT_NV; ++k) {                                                               \
        const int i = tid + LT_THREADS * (k + LT_NV * (bt));                                                          \
        const int rr = i >> sh_q, c4 = i & (q4 - 1);                                                                  \
        if (rr < nrow) {                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                if (4 * c4 + j < NC + 2) gS[rr * SW + 4 * c4 + j] = lv[k][j];                                         \
        }                                                                                                             \
    }
    if (a.ld) LT_LOAD_BATCH(0);

    float cfg_ = 0.f, ckeep = 0.f;
    {                                   // every workgroup counts all labels: two per 16-byte load, 8 loads in flight
        typedef long long lt_v2l __attribute__((ext_vector_type(2)));
        typedef const lt_v2l __attribute__((address_space(1))) *lt_gv2lp;
        const int R2 = R >> 1;
        // LT_CNT loads per thread requested at once (clamped index, masked sum): B*L2 = 4 480 labels are ONE batch of 128 threads
        // x 18 (three batches of 8 before: three dependent round trips on the critical path of every workgroup)
        constexpr int LT_CNT = 18;
        for (int base = 0; base < R2; base += LT_CNT * LT_THREADS) {
            lt_v2l lab[LT_CNT];
#pragma unroll
            for (int k = 0; k < LT_CNT; ++k) lab[k] = *(lt_gv2lp)(a.cls_label + 2 * min(base + tid + k * LT_THREADS, R2 - 1));
#pragma unroll
            for (int k = 0; k < LT_CNT; ++k) {
                const bool in = base + tid + k * LT_THREADS < R2;
                cfg_ += in ? ((lab[k][0] == 1) ? 1.f : 0.f) + ((lab[k][1] == 1) ? 1.f : 0.f) : 0.f;
                ckeep += in ? ((lab[k][0] != -1) ? 1.f : 0.f) + ((lab[k][1] != -1) ? 1.f : 0.f) : 0.f;
            }
        }
        if ((R & 1) && tid == 0) {
            const int64_t lab = a.cls_label[R - 1];
            cfg_ += (lab == 1) ? 1.f : 0.f;
            ckeep += (lab != -1) ? 1.f : 0.f;
        }
    }
    float cnt2[2] = {cfg_, ckeep};
    block_sums<LT_THREADS, 2>(cnt2, shN);          // (sums of 0 / 1: exact in any order)
    const float nfg = cnt2[0], nkeep = cnt2[1];
    const float inv_cls = 1.f / (nfg + 1e-14f);
    // a batch without a foreground row (the reference asserts on it, det_base.py:416): the foreground means report 0
    // instead of 0 * inf = NaN, and out[11] = nfg lets the caller see it without a host sync on the hot path
    const float inv_fg = nfg > 0.f ? 1.f / nfg : 0.f;

    float acc[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) acc[i] = 0.f;

    if (a.ld) {
        LT_STAGE_BATCH(0);
        if (nbatch > 1) {
            LT_LOAD_BATCH(1);
            LT_STAGE_BATCH(1);
        }
        __syncthreads();
    }

    for (int r = blockIdx.x * LT_THREADS + tid; r < R; r += gridDim.x * LT_THREADS) {
        const int b = r / L2, l = r % L2;
        const int64_t lab = a.cls_label[r];
        // ---------------- focal classification loss (common.py:217-232)
        const float c0 = a.ld ? gS[tid * SW] : a.cls_raw[((int64_t)b * 2 + 0) * L2 + l];
        const float c1 = a.ld ? gS[tid * SW + 1] : a.cls_raw[((int64_t)b * 2 + 1) * L2 + l];
        const float m = fmaxf(c0, c1);
        const float e0 = expf(c0 - m), e1 = expf(c1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        float g0 = 0.f, g1 = 0.f;
        if (lab != -1) {
            const int t = lab >= 1 ? 1 : 0;
            const float pt = t ? p1 : p0;
            const float alpha = t ? 0.25f : 0.75f;
            const float om = 1.f - pt, lg = logf(pt + 1e-14f);
            acc[1] += -alpha * om * om * lg;
            const float dpt = -alpha * (-2.f * om * lg + om * om / (pt + 1e-14f));
            const float gt_ = dpt * pt * om * inv_cls;
            g0 = t ? -gt_ : gt_;
            g1 = t ? gt_ : -gt_;
            acc[8] += ((p1 > p0 ? 1 : 0) == t) ? 1.f : 0.f;
        }
        if (a.dcls) {
            if (a.ld) {
                gG[tid * SW] = g0;
                gG[tid * SW + 1] = g1;
            } else {
                a.dcls[((int64_t)b * 2 + 0) * L2 + l] = g0;
                a.dcls[((int64_t)b * 2 + 1) * L2 + l] = g1;
            }
        }
        float *go = gG + tid * SW + 2;
#pragma unroll
        for (int j = 0; j < NC; ++j) go[j] = 0.f;
        if (lab == 1) {
            if (!a.ld) {
#pragma unroll
                for (int j = 0; j < NC; ++j) gS[tid * SW + 2 + j] = a.reg_raw[((int64_t)b * NC + j) * L2 + l];
            }
            const float *o = gS + tid * SW + 2;
            const float rx = a.ref2[((int64_t)b * 3 + 0) * L2 + l], ry = a.ref2[((int64_t)b * 3 + 1) * L2 + l],
                        rz = a.ref2[((int64_t)b * 3 + 2) * L2 + l];
            const float clx = a.box_center[b * 3], cly = a.box_center[b * 3 + 1], clz = a.box_center[b * 3 + 2];
            const float hlab = a.box_heading[b];
            const float sl0 = a.box_size[b * 3], sl1 = a.box_size[b * 3 + 1], sl2 = a.box_size[b * 3 + 2];
            const int sc = min(max((int)a.size_class[b], 0), NS - 1);     // clamped: an out-of-range label must not index out of bounds
            const float ex0 = a.mean_size[sc * 3], ex1 = a.mean_size[sc * 3 + 1], ex2 = a.mean_size[sc * 3 + 2];
            const float wB = a.w_box * inv_fg;

            // ---- centre (huber on the distance, delta 3)
            {
                const float dx = clx - rx - o[0], dy = cly - ry - o[1], dz = clz - rz - o[2];
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                const float q = fminf(dist, 3.f);
                acc[2] += 0.5f * q * q + 3.f * (dist - q);
                if (dist > 0.f) {
                    const float k = -wB * q / dist;
                    go[0] += k * dx; go[1] += k * dy; go[2] += k * dz;
                }
            }
            // ---- heading: class CE + residual huber (box_transform.py:55-65)
            const float ga = pymod(hlab, TWO_PI);
            const float shifted = pymod(ga + half, TWO_PI);
            int hc = (int)floorf(shifted / per);
            hc = hc < 0 ? 0 : (hc > NB - 1 ? NB - 1 : hc);
            const float hres = (shifted - ((float)hc * per + half)) / half;
            {
                float mx = o[3];
                int am = 0;
#pragma unroll
                for (int j = 1; j < NB; ++j)
                    if (o[3 + j] > mx) { mx = o[3 + j]; am = j; }
                float se = 0.f, ej[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) { ej[j] = expf(o[3 + j] - mx); se += ej[j]; }
                acc[3] += logf(se) + mx - o[3 + hc];
#pragma unroll
                for (int j = 0; j < NB; ++j) go[3 + j] += wB * (ej[j] / se - (j == hc ? 1.f : 0.f));
                acc[9] += (am == hc) ? 1.f : 0.f;
                const float e = o[3 + NB + hc] - hres;
                const float ab = fabsf(e), q = fminf(ab, 1.f);
                acc[4] += 0.5f * q * q + (ab - q);
                go[3 + NB + hc] += wB * a.w_headreg * (e > 0.f ? q : (e < 0.f ? -q : 0.f));
            }
            // ---- size: class CE + residual huber on the norm (box_transform.py:5-19)
            {
                const float *ss = &o[3 + 2 * NB];
                float mx = ss[0];
                int am = 0;
#pragma unroll
                for (int j = 1; j < NS; ++j)
                    if (ss[j] > mx) { mx = ss[j]; am = j; }
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < NS; ++j) {          // e_j parked in the gradient row (LDS), not in a private array
                    const float ej = expf(ss[j] - mx);
                    go[3 + 2 * NB + j] = ej;
                    se += ej;
                }
                acc[5] += logf(se) + mx - ss[sc];
#pragma unroll
                for (int j = 0; j < NS; ++j) go[3 + 2 * NB + j] = wB * (go[3 + 2 * NB + j] / se - (j == sc ? 1.f : 0.f));
                acc[10] += (am == sc) ? 1.f : 0.f;
            }
            const int so = 3 + 2 * NB + NS + sc * 3;      // the selected size-residual triple
            {
                const float v0 = (sl0 - ex0) / ex0 - o[so], v1 = (sl1 - ex1) / ex1 - o[so + 1],
                            v2 = (sl2 - ex2) / ex2 - o[so + 2];
                const float n = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
                const float q = fminf(n, 1.f);
                acc[6] += 0.5f * q * q + (n - q);
                if (n > 0.f) {
                    const float k = -wB * a.w_sizereg * q / n;
                    go[so] += k * v0; go[so + 1] += k * v1; go[so + 2] += k * v2;
                }
            }
            // ---- corner loss (model_util.py:48-72, det_base.py:314-332)
            {
                float ang = (float)hc * per + o[3 + NB + hc] * half;
                if (ang > PI) ang -= TWO_PI;
                const float pl = o[so] * ex0 + ex0, pw = o[so + 1] * ex1 + ex1, ph = o[so + 2] * ex2 + ex2;
                const float pcx = rx + o[0], pcy = ry + o[1], pcz = rz + o[2];
                float px[8], py[8], pz[8], gx[8], gy[8], gz[8], fx[8], fy[8], fz[8];
                // one cos / sin pair per angle (the precise fp32 routines are the long pole of this lane: the row's work is
                // serial); the flipped label box takes cos / sin of hlab + pi evaluated like the reference does
                const float c = cosf(ang), s = sinf(ang);
                const float hf = hlab + PI;
                corners8(pcx, pcy, pcz, c, s, pl, pw, ph, px, py, pz);
                corners8(clx, cly, clz, cosf(hlab), sinf(hlab), sl0, sl1, sl2, gx, gy, gz);
                corners8(clx, cly, clz, cosf(hf), sinf(hf), sl0, sl1, sl2, fx, fy, fz);
                float d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    d1 += sqrtf((px[k] - gx[k]) * (px[k] - gx[k]) + (py[k] - gy[k]) * (py[k] - gy[k]) +
                                (pz[k] - gz[k]) * (pz[k] - gz[k]));
                    d2 += sqrtf((px[k] - fx[k]) * (px[k] - fx[k]) + (py[k] - fy[k]) * (py[k] - fy[k]) +
                                (pz[k] - fz[k]) * (pz[k] - fz[k]));
                }
                d1 *= 0.125f; d2 *= 0.125f;
                const bool first = d1 <= d2;
                const float cd = first ? d1 : d2;
                const float q = fminf(cd, 1.f);
                acc[7] += 0.5f * q * q + (cd - q);
                const float wk = wB * a.w_corner * q * 0.125f;
                float dcx = 0.f, dcy = 0.f, dcz = 0.f, dl = 0.f, dw = 0.f, dh = 0.f, dang = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float tx = first ? gx[k] : fx[k], ty = first ? gy[k] : fy[k], tz = first ? gz[k] : fz[k];
                    const float ex_ = px[k] - tx, ey_ = py[k] - ty, ez_ = pz[k] - tz;
                    const float nn = sqrtf(ex_ * ex_ + ey_ * ey_ + ez_ * ez_);
                    if (nn > 0.f) {
                        const float kx = wk * ex_ / nn, ky = wk * ey_ / nn, kz = wk * ez_ / nn;
                        const float sx = (k & 2) ? -0.5f : 0.5f, sy = (k & 4) ? -0.5f : 0.5f;
                        const float sz = ((k & 3) == 1 || (k & 3) == 2) ? -0.5f : 0.5f;
                        const float x = sx * pl, z = sz * pw;
                        dcx += kx; dcy += ky; dcz += kz;
                        dl += kx * c * sx - kz * s * sx;
                        dw += kx * s * sz + kz * c * sz;
                        dh += ky * sy;
                        dang += kx * (-s * x + c * z) + kz * (-c * x - s * z);
                    }
                }
                go[0] += dcx; go[1] += dcy; go[2] += dcz;
                go[so] += dl * ex0; go[so + 1] += dw * ex1; go[so + 2] += dh * ex2;
                go[3 + NB + hc] += dang * half;
            }
        }
        if (a.ld) {
            // (the gradient row already sits in gG)
        } else if (a.dreg) {
#pragma unroll
            for (int j = 0; j < NC; ++j) a.dreg[((int64_t)b * NC + j) * L2 + l] = go[j];
        }
    }
    if (a.ld && a.dcls) {              // (the row loop runs at most once per thread: the grid covers R)
        __syncthreads();
        for (int i = tid; i < nrow * q4; i += LT_THREADS) {        // 16-byte stores: a quarter of the store instructions
            const int rr = i >> sh_q, c4 = i & (q4 - 1);
            lt_v4f v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 4 * c4 + j < NC + 2 ? gG[rr * SW + 4 * c4 + j] : 0.f;
            *(lt_v4f *)(a.dcls + (int64_t)(row0 + rr) * a.ld + 4 * c4) = v;
        }
    }
    // ---- combine the workgroups: out[1..10] accumulate, out[15] (as int) is the arrival ticket; both were zeroed by the
    // hipMemsetAsync in front of the launch.  Accumulators are written and read with device-scope atomics only.
    // (one barrier pair for the ten sums, each still wave sum first, then the waves in order: the values of ten block_sum calls)
    float tot[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) tot[i] = acc[i];
    block_sums<LT_THREADS, 11>(tot, shN);
    if (tid == 0) {
        int ticket;
        if (a.scratch) {
#pragma unroll
            for (int i = 1; i < 11; ++i) a.scratch[32 + 16 * blockIdx.x + i] = tot[i];
            __threadfence();
            ticket = atomicAdd((int *)a.scratch, 1);
        } else {
#pragma unroll
            for (int i = 1; i < 11; ++i) atomicAdd(&a.out[i], tot[i]);
            __threadfence();
            ticket = atomicAdd((int *)&a.out[15], 1);
        }
        last_s = (ticket == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    // The last workgroup sums the partials.  With the scratch buffer: IN WORKGROUP ORDER (reproducible scalars), but not by
    // one thread walking gridDim.x x 10 dependent loads (35 memory round trips on the critical path): thread g fetches the
    // ten partials of workgroups g, g + T, ... (independent loads, summed in that fixed order), then thread 0 adds the
    // per-thread sums in thread order from LDS.
    if (last_s && a.scratch) {
        __threadfence();
        float pt_[11];
#pragma unroll
        for (int i = 1; i < 11; ++i) pt_[i] = 0.f;
        for (int g = tid; g < (int)gridDim.x; g += LT_THREADS)
#pragma unroll
            for (int i = 1; i < 11; ++i)
                pt_[i] += __hip_atomic_load(&a.scratch[32 + 16 * g + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 1; i < 11; ++i) gG[tid * 11 + i] = pt_[i];        // (gG is free: the gradient rows have left)
    }
    __syncthreads();
    if (last_s && tid == 0) {
        float t[11];
        if (a.scratch) {
#pragma unroll
            for (int i = 1; i < 11; ++i) t[i] = 0.f;
            const int nth = min((int)gridDim.x, LT_THREADS);
            for (int g = 0; g < nth; ++g)
#pragma unroll
                for (int i = 1; i < 11; ++i) t[i] += gG[g * 11 + i];
            ((int *)a.scratch)[0] = 0;          // ready for the next launch
        } else {
            __threadfence();
#pragma unroll
            for (int i = 1; i < 11; ++i) t[i] = __hip_atomic_load(&a.out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float cls = t[1] * inv_cls;
        const float center = t[2] * inv_fg, hcls = t[3] * inv_fg, hres = t[4] * inv_fg;
        const float scls = t[5] * inv_fg, sres = t[6] * inv_fg, corner = t[7] * inv_fg;
        const float total = cls + a.w_box * (center + hcls + scls + a.w_headreg * hres + a.w_sizereg * sres + a.w_corner * corner);
        a.out[0] = total;
        a.out[1] = cls; a.out[2] = center; a.out[3] = hcls; a.out[4] = hres;
        a.out[5] = scls; a.out[6] = sres; a.out[7] = corner;
        a.out[8] = nkeep > 0.f ? t[8] / nkeep : 0.f; a.out[9] = t[9] * inv_fg; a.out[10] = t[10] * inv_fg;
        a.out[11] = nfg;
        a.out[12] = a.out[13] = a.out[14] = a.out[15] = 0.f;
        if (a.total) a.total[0] = total;
    }
}

static int launch_loss(const LossArgs &a, int ns, hipStream_t st)
{
    if (!a.scratch) {                   // accumulate-into-out path: the accumulators and the ticket start from zero
        hipError_t e = hipMemsetAsync(a.out, 0, 16 * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    const int R = a.B * a.L2;
    if (ns == 3) hipLaunchKernelGGL((loss_tail_kernel<3, 128>), dim3((R + 127) / 128), dim3(128), 0, st, a);
    else hipLaunchKernelGGL((loss_tail_kernel<10, 64>), dim3((R + 63) / 64), dim3(64), 0, st, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

extern "C" int fcn_det_loss_tail(const float *cls_raw, const float *reg_raw, const int64_t *cls_label,
                                 const float *center_ref2, const float *box3d_center, const float *box3d_heading,
                                 const float *box3d_size, const int64_t *size_class, const float *mean_size,
                                 int B, int L2, int num_heading_bin, int num_size_cluster,
                                 float w_box, float w_corner, float w_headreg, float w_sizereg,
                                 float *out16, float *dcls, float *dreg, void *stream)
{
    if (!cls_raw || !reg_raw || !cls_label || !center_ref2 || !box3d_center || !box3d_heading || !box3d_size ||
        !size_class || !mean_size || !out16)
        return FCN_E_BADARG;
    if (num_heading_bin != LT_NB || (num_size_cluster != 3 && num_size_cluster != 10)) return FCN_E_LIMIT;
    if (B <= 0 || L2 <= 0) return FCN_E_BADARG;
    if ((uintptr_t)cls_label & 15) return FCN_E_BADARG;          // the label count reads two int64 per 16-byte load
    LossArgs a;
    a.cls_raw = cls_raw; a.reg_raw = reg_raw; a.cls_label = cls_label; a.ref2 = center_ref2;
    a.box_center = box3d_center; a.box_heading = box3d_heading; a.box_size = box3d_size; a.size_class = size_class;
    a.mean_size = mean_size; a.out = out16; a.dcls = dcls; a.dreg = dreg; a.B = B; a.L2 = L2;
    a.w_box = w_box; a.w_corner = w_corner; a.w_headreg = w_headreg; a.w_sizereg = w_sizereg; a.ld = 0;
    a.scratch = nullptr; a.total = nullptr;
    return launch_loss(a, num_size_cluster, (hipStream_t)stream);
}

extern "C" int fcn_det_loss_tail_rows2(const float *logits, const int64_t *cls_label, const float *center_ref2,
                                       const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                                       const int64_t *size_class, const float *mean_size, int B, int L2,
                                       int num_heading_bin, int num_size_cluster,
                                       float w_box, float w_corner, float w_headreg, float w_sizereg,
                                       float *out16, float *dlogits, float *scratch, float *total, void *stream);

extern "C" int fcn_det_loss_tail_rows(const float *logits, const int64_t *cls_label, const float *center_ref2,
                                      const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                                      const int64_t *size_class, const float *mean_size, int B, int L2,
                                      int num_heading_bin, int num_size_cluster,
                                      float w_box, float w_corner, float w_headreg, float w_sizereg,
                                      float *out16, float *dlogits, void *stream)
{
    return fcn_det_loss_tail_rows2(logits, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size, size_class,
                                   mean_size, B, L2, num_heading_bin, num_size_cluster, w_box, w_corner, w_headreg,
                                   w_sizereg, out16, dlogits, nullptr, nullptr, stream);
}

extern "C" int fcn_det_loss_tail_scratch_floats(int B, int L2)
{
    return (B <= 0 || L2 <= 0) ? 0 : 32 + 16 * ((B * L2 + LT_MIN_THREADS - 1) / LT_MIN_THREADS);   // (covers both workgroup sizes)
}

extern "C" int fcn_det_loss_tail_rows2(const float *logits, const int64_t *cls_label, const float *center_ref2,
                                       const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                                       const int64_t *size_class, const float *mean_size, int B, int L2,
                                       int num_heading_bin, int num_size_cluster,
                                       float w_box, float w_corner, float w_headreg, float w_sizereg,
                                       float *out16, float *dlogits, float *scratch, float *total, void *stream)
{
    if (!logits || !cls_label || !center_ref2 || !box3d_center || !box3d_heading || !box3d_size || !size_class ||
        !mean_size || !out16)
        return FCN_E_BADARG;
    if (num_heading_bin != LT_NB || (num_size_cluster != 3 && num_size_cluster != 10)) return FCN_E_LIMIT;
    if (B <= 0 || L2 <= 0) return FCN_E_BADARG;
    // 16-byte vector accesses: two int64 labels per load, the logits rows as float4 loads, the gradient rows as float4 stores
    if (((uintptr_t)cls_label & 15) || ((uintptr_t)logits & 15) || ((uintptr_t)dlogits & 15)) return FCN_E_BADARG;
    LossArgs a;
    a.cls_raw = logits; a.reg_raw = nullptr; a.cls_label = cls_label; a.ref2 = center_ref2;
    a.box_center = box3d_center; a.box_heading = box3d_heading; a.box_size = box3d_size; a.size_class = size_class;
    a.mean_size = mean_size; a.out = out16; a.dcls = dlogits; a.dreg = nullptr; a.B = B; a.L2 = L2;
    a.w_box = w_box; a.w_corner = w_corner; a.w_headreg = w_headreg; a.w_sizereg = w_sizereg;
    a.ld = (2 + 3 + 2 * num_heading_bin + 4 * num_size_cluster <= 64) ? 64 : 128;
    a.scratch = scratch; a.total = total;
    return launch_loss(a, num_size_cluster, (hipStream_t)stream);
}
