// ROUND-2 SCOUTING, not part of libfcn_hip.so and not on the product path: conv3 of one PointNet scale
// (Y = relu(bn2(y2)) . W3^T over the live entry rows) with bf16 MFMA operands and fp32 accumulation, to measure what a
// bf16 throughput mode buys on the dominant GEMM and what it costs in accuracy.  Same tile
// list / workspace as csrc/pointnet_fwd.hip's fwd_gemm_kernel<1,2,2>; operands are converted fp32 -> bf16 (RNE) while
// they are staged in LDS (row-major, k contiguous: one ds_read_b128 per 32x32x16 operand).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float v4f __attribute__((ext_vector_type(4)));

#define KC 32
#define LDS_ROW 40            // bf16 elements per staged row (32 + 8 padding: 80-byte rows)

__device__ __forceinline__ unsigned short f2bf(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);          // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ int acc_row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }

struct ScoutArgs {
    const int32_t *woff, *tiles;
    const float *aprev, *bn_in, *W;
    float *y;
    int L, cap, CIN, COUT, tps;
};

__global__ __launch_bounds__(256) void conv3_bf16_kernel(ScoutArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned short As[128 * LDS_ROW];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[128 * LDS_ROW];
    __shared__ float sS[512], tS[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    if ((int)blockIdx.x >= a.tiles[0]) return;
    const int code = a.tiles[4 + blockIdx.x];
    const int b = code / a.tps, t = code % a.tps;
    const int nent = a.woff[(int64_t)b * (a.L + 1) + a.L];
    const int row0 = t * 128;
    const int nvalid = min(128, nent - row0);
    const int64_t grow0 = (int64_t)b * a.cap + row0;
    const int n0 = blockIdx.y * 128;
    const int CIN = a.CIN, COUT = a.COUT;
    for (int i = tid; i < CIN; i += 256) { sS[i] = a.bn_in[i]; tS[i] = a.bn_in[CIN + i]; }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    v4f ra[4], rw[4];
    const int nchunk = CIN / KC;
#define SC_LOAD(cc)                                                                                   \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;                                      \
            const int rc = min(r, nvalid - 1);                                                        \
            ra[i] = *(const v4f *)(a.aprev + (grow0 + rc) * CIN + (cc) * KC + 4 * kq);                \
            rw[i] = *(const v4f *)(a.W + (int64_t)(n0 + r) * CIN + (cc) * KC + 4 * kq);               \
        }                                                                                             \
    }
    SC_LOAD(0);
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            const bool ok = r < nvalid;
            const float *sp = sS + c * KC + 4 * kq, *tp = tS + c * KC + 4 * kq;
            const float v0 = ok ? fmaxf(fmaf(sp[0], ra[i].x, tp[0]), 0.f) : 0.f;
            const float v1 = ok ? fmaxf(fmaf(sp[1], ra[i].y, tp[1]), 0.f) : 0.f;
            const float v2 = ok ? fmaxf(fmaf(sp[2], ra[i].z, tp[2]), 0.f) : 0.f;
            const float v3 = ok ? fmaxf(fmaf(sp[3], ra[i].w, tp[3]), 0.f) : 0.f;
            uint2 pa, pb;
            pa.x = f2bf(v0) | ((unsigned)f2bf(v1) << 16); pa.y = f2bf(v2) | ((unsigned)f2bf(v3) << 16);
            pb.x = f2bf(rw[i].x) | ((unsigned)f2bf(rw[i].y) << 16); pb.y = f2bf(rw[i].z) | ((unsigned)f2bf(rw[i].w) << 16);
            *(uint2 *)(As + r * LDS_ROW + 4 * kq) = pa;
            *(uint2 *)(Bs + r * LDS_ROW + 4 * kq) = pb;
        }
        __syncthreads();
        if (c + 1 < nchunk) SC_LOAD(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = *(const bf16x8 *)(As + (wm * 64 + i * 32 + l31) * LDS_ROW + ks * 16 + 8 * lh);
                bv[i] = *(const bf16x8 *)(Bs + (wn * 64 + i * 32 + l31) * LDS_ROW + ks * 16 + 8 * lh);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wm * 64 + mt * 32 + acc_row(reg, lh);
                const int col = n0 + wn * 64 + nt * 32 + l31;
                if (row < nvalid) a.y[(grow0 + row) * COUT + col] = acc[mt][nt][reg];
            }
}

extern "C" int scout_conv3_bf16(const int32_t *woff, const int32_t *tiles, const float *y2, const float *bn2, const float *W3,
                                float *y3_out, int B, int L, int K, int C2, int C3, void *stream)
{
    if (C2 % 32 || C3 % 128 || C2 > 512) return 1;
    ScoutArgs a;
    a.woff = woff; a.tiles = tiles; a.aprev = y2; a.bn_in = bn2; a.W = W3; a.y = y3_out;
    a.L = L; a.cap = L * K; a.CIN = C2; a.COUT = C3; a.tps = (a.cap + 127) / 128;
    hipLaunchKernelGGL(conv3_bf16_kernel, dim3(B * a.tps, C3 / 128), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
