// Stand-alone micro-benchmark (hipcc --offload-arch=gfx950 -O3 -I frustum_convnet_amd/csrc tools/micro/pgemm.hip -o tools/micro/pgemm):
// the CEILING of a PointNet layer-3 GEMM of the widest scale as a PURE GEMM -- Y[M,N] = A[M,K] . W[N,K]^T with three
// v_mfma_f32_32x32x16_f16 per product (the fp16 x 3 split), both operands PRE-ENCODED in the kb-major images of gemm_tile.h, so that the
// K loop holds no arithmetic but the MFMAs: M = 36 363, K = 256, N = 512 (conv3 forward / its data gradient with K and N swapped).
//   reg   : operands staged global -> VGPR -> LDS (16-byte copies), one LDS buffer, two barriers per chunk -- the product kernels' loop
//           without their BatchNorm / ReLU / split-encode VALU
//   glds2 : both operands by LDS-DMA (__builtin_amdgcn_global_load_lds, 16 bytes per lane), two LDS buffers, one barrier per chunk
//   glds3 : three LDS buffers, two chunks in flight, raw s_barrier + counted vmcnt (cdna_hip_programming.md, glds rules)
// each with and without the output epilogue (74 MB of fp32 stores through the transposition patch), plus the one-pass ENCODE kernel
// that would have to produce the A image from y2 (BatchNorm + ReLU + split, 37 MB in, 37 MB out).  Results are checked against a
// host fp64 product on a sample of outputs.  Not part of the product.
#include "gemm_tile.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const void __attribute__((address_space(1))) *gvoidp;
typedef void __attribute__((address_space(3))) *lvoidp;

constexpr int TM = 128, TN = 128, MT = 2, NT = 2, NTHR = 256;
constexpr int LDRA = KbTile<TM>::LDR, LDRB = KbTile<TN>::LDR;
constexpr int CH_U4 = KbTile<TM>::U4 + KbTile<TN>::U4;          // u32x4 of one chunk stage (A image + W image)

// chunk c of the A image: [2 planes][4 k-blocks][Mp rows] u32x4 at Aimg + c * 8 * Mp; of the W image: [2][4][N] at Wimg + c * 8 * N
template <int MODE, int STORE>
__global__ __launch_bounds__(NTHR) void pg_kernel(const u32x4 *__restrict__ Aimg, const u32x4 *__restrict__ Wimg, float *__restrict__ Y,
                                                  int M, int Mp, int N, int K)
{
    constexpr int NBUF = MODE == 0 ? 1 : (MODE == 1 ? 2 : 3);
    __shared__ u32x4 lds4[NBUF * CH_U4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ny = N / TN;
    const int nrt = (M + TM - 1) / TM;
    const int xt = fcn_xcd_tile(blockIdx.x, nrt * ny);
    if (xt < 0) return;
    const int bx = xt / ny, by = xt % ny;
    const int row0 = bx * TM, n0 = by * TN;
    const int nchunk = K / KC;
    f32x16 acc[MT][NT];
    acc_zero<MT, NT>(acc);

    if constexpr (MODE == 0) {
        u32x4 *Ab = lds4, *Bb = lds4 + KbTile<TM>::U4;
        u32x4 ra[4], rw[4];
        auto load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + NTHR * i;                  // (plane, kb) row f / 128, column f % 128
                ra[i] = ldgu4(Aimg + ((int64_t)c * 8 + f / TM) * Mp + row0 + f % TM);
                rw[i] = ldgu4(Wimg + ((int64_t)c * 8 + f / TN) * N + n0 + f % TN);
            }
        };
        load(0);
        for (int c = 0; c < nchunk; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + NTHR * i;
                Ab[(f / TM) * LDRA + f % TM] = ra[i];
                Bb[(f / TN) * LDRB + f % TN] = rw[i];
            }
            __syncthreads();
            if (c + 1 < nchunk) load(c + 1);
            mma_chunk_kb<MM_F16X3, MT, NT, LDRA, LDRB>(Ab, Bb, wm * 32 * MT, wn * 32 * NT, acc);
            __syncthreads();
        }
    } else {
        // LDS-DMA: one instruction moves 64 lanes x 16 bytes to a wave-uniform LDS base + lane * 16 -- a 64-row run of one (plane, kb)
        // row of an image.  A chunk = 16 runs of A + 16 runs of W; wave w issues runs 8w .. 8w+7.
        auto issue = [&](int c, int buf) __attribute__((always_inline)) {
            u32x4 *Ab = lds4 + buf * CH_U4, *Bb = Ab + KbTile<TM>::U4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int run = wave * 8 + i;                  // 0..15 A, 16..31 W
                const int r = run & 15, pk = r >> 1, half = r & 1;
                if (run < 16) {
                    const u32x4 *g = Aimg + ((int64_t)c * 8 + pk) * Mp + row0 + half * 64 + lane;
                    __builtin_amdgcn_global_load_lds((gvoidp)g, (lvoidp)(Ab + pk * LDRA + half * 64), 16, 0, 0);
                } else {
                    const u32x4 *g = Wimg + ((int64_t)c * 8 + pk) * N + n0 + half * 64 + lane;
                    __builtin_amdgcn_global_load_lds((gvoidp)g, (lvoidp)(Bb + pk * LDRB + half * 64), 16, 0, 0);
                }
            }
        };
        if constexpr (MODE == 1) {
            issue(0, 0);
            for (int c = 0; c < nchunk; ++c) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                // chunk c landed for every wave; buffer (c+1)&1 is free (its MFMAs ran before this barrier)
                if (c + 1 < nchunk) issue(c + 1, (c + 1) & 1);
                const u32x4 *Ab = lds4 + (c & 1) * CH_U4;
                mma_chunk_kb<MM_F16X3, MT, NT, LDRA, LDRB>(Ab, Ab + KbTile<TM>::U4, wm * 32 * MT, wn * 32 * NT, acc);
            }
        } else {
            issue(0, 0);
            if (nchunk > 1) issue(1, 1);
            for (int c = 0; c < nchunk; ++c) {
                // 8 DMA instructions per chunk and wave: leave the newest chunk in flight
                if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (c + 2 < nchunk) issue(c + 2, (c + 2) % 3);
                const u32x4 *Ab = lds4 + (c % 3) * CH_U4;
                mma_chunk_kb<MM_F16X3, MT, NT, LDRA, LDRB>(Ab, Ab + KbTile<TM>::U4, wm * 32 * MT, wn * 32 * NT, acc);
            }
        }
        __syncthreads();
    }
    if constexpr (STORE) {
        float *patch = (float *)lds4 + wave * EP_FLOATS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ep_put(patch, acc[mt][nt], l31, lh);
                __builtin_amdgcn_wave_barrier();
                const int rbase = row0 + wm * 32 * MT + mt * 32, cbase = n0 + wn * 32 * NT + nt * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = lane + 64 * q, row = rbase + (idx >> 3);
                    const v4f v = ep_get(patch, lane, q);
                    if (row < M) sts4(Y + (int64_t)row * N + cbase + 4 * (idx & 7), v);
                }
                __builtin_amdgcn_wave_barrier();
            }
    } else {
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[mt][nt][r];
        if (s == 123.456f) Y[0] = s;
    }
}

// one-pass encode: A image of relu(s * y + t), y (M, K) fp32 rows.  A workgroup = 64 rows; thread -> (row, 32-byte k-block) pieces.
__global__ __launch_bounds__(256) void enc_kernel(const float *__restrict__ y, const float *__restrict__ sc, const float *__restrict__ sh,
                                                  u32x4 *__restrict__ Aimg, int M, int Mp, int K)
{
    __shared__ u32x4 img[2 * 4 * 66];
    const int tid = threadIdx.x, row0 = blockIdx.x * 64;
    for (int c = 0; c < K / KC; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            const int row = min(row0 + r, M - 1);
            const v4f v = ldg4(y + (int64_t)row * K + c * KC + 4 * kq);
            const v4f s4 = ldg4(sc + c * KC + 4 * kq), t4 = ldg4(sh + c * KC + 4 * kq);
            const bool ok = row0 + r < M;
            kb_store4<MM_F16X3, 66>(img, r, kq, ok ? fmaxf(fmaf(s4.x, v.x, t4.x), 0.f) : 0.f, ok ? fmaxf(fmaf(s4.y, v.y, t4.y), 0.f) : 0.f,
                                    ok ? fmaxf(fmaf(s4.z, v.z, t4.z), 0.f) : 0.f, ok ? fmaxf(fmaf(s4.w, v.w, t4.w), 0.f) : 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, pk = f >> 6, r = f & 63;
            Aimg[((int64_t)c * 8 + pk) * Mp + row0 + r] = img[pk * 66 + r];
        }
        __syncthreads();
    }
}

static void host_encode(const std::vector<float> &x, int rows, int rowsp, int K, std::vector<uint32_t> &img)
{
    // kb-major image, fp16 split: u32x4 (c, plane, kb, row) = packed parts of x[row][c*32 + kb*8 + 0..7]
    img.assign((size_t)(K / 32) * 8 * rowsp * 4, 0u);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; k += 2) {
            const float x0 = x[(size_t)r * K + k], x1 = x[(size_t)r * K + k + 1];
            const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
            const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
            const uint16_t b[4] = {__builtin_bit_cast(uint16_t, h0), __builtin_bit_cast(uint16_t, h1), __builtin_bit_cast(uint16_t, l0),
                                   __builtin_bit_cast(uint16_t, l1)};
            const int c = k / 32, kb = (k % 32) / 8, q = (k % 8) / 2;
            img[((((size_t)c * 2 + 0) * 4 + kb) * rowsp + r) * 4 + q] = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
            img[((((size_t)c * 2 + 1) * 4 + kb) * rowsp + r) * 4 + q] = (uint32_t)b[2] | ((uint32_t)b[3] << 16);
        }
}

template <int MODE, int STORE>
static float run(const u32x4 *A, const u32x4 *W, float *Y, int M, int Mp, int N, int K, int iters)
{
    const int nt = ((M + TM - 1) / TM) * (N / TN);
    const dim3 grid((nt + 7) / 8 * 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pg_kernel<MODE, STORE>), grid, dim3(NTHR), 0, 0, A, W, Y, M, Mp, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pg_kernel<MODE, STORE>), grid, dim3(NTHR), 0, 0, A, W, Y, M, Mp, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 36363, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 512;
    const int Mp = (M + 127) / 128 * 128;
    std::vector<float> a((size_t)M * K), w((size_t)N * K);
    srand(7);
    for (auto &v : a) { v = (float)rand() / RAND_MAX * 2.f - 0.5f; if (v < 0.f) v = 0.f; }       // relu-like activations
    for (auto &v : w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    std::vector<uint32_t> ai, wi;
    host_encode(a, M, Mp, K, ai);
    host_encode(w, N, N, K, wi);
    u32x4 *dA, *dW, *dA2;
    float *dY, *dy2, *dsc, *dsh;
    CK(hipMalloc(&dA, ai.size() * 4)); CK(hipMalloc(&dW, wi.size() * 4)); CK(hipMalloc(&dA2, ai.size() * 4));
    CK(hipMalloc(&dY, (size_t)M * N * 4)); CK(hipMalloc(&dy2, (size_t)M * K * 4));
    CK(hipMalloc(&dsc, K * 4)); CK(hipMalloc(&dsh, K * 4));
    CK(hipMemcpy(dA, ai.data(), ai.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, wi.data(), wi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy2, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ones(K, 1.f), zeros(K, 0.f);
    CK(hipMemcpy(dsc, ones.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsh, zeros.data(), K * 4, hipMemcpyHostToDevice));
    const double gflop = 2.0 * M * K * N * 1e-9;
    printf("pure GEMM  M=%d K=%d N=%d  (%.2f GFLOP useful, x3 MFMA issued; y out %.1f MB, A in %.1f MB)\n", M, K, N, gflop,
           (double)M * N * 4e-6, (double)M * K * 4e-6);
    const int it = 50;
    struct { const char *name; float us; } res[6];
    res[0] = {"reg   no-store", run<0, 0>(dA, dW, dY, M, Mp, N, K, it)};
    res[1] = {"reg   store   ", run<0, 1>(dA, dW, dY, M, Mp, N, K, it)};
    res[2] = {"glds2 no-store", run<1, 0>(dA, dW, dY, M, Mp, N, K, it)};
    res[3] = {"glds2 store   ", run<1, 1>(dA, dW, dY, M, Mp, N, K, it)};
    res[4] = {"glds3 no-store", run<2, 0>(dA, dW, dY, M, Mp, N, K, it)};
    res[5] = {"glds3 store   ", run<2, 1>(dA, dW, dY, M, Mp, N, K, it)};
    for (auto &r : res) printf("  %s  %8.2f us   %7.1f TFLOP/s useful  %7.1f issued\n", r.name, r.us, gflop / r.us * 1e-3 * 1e3, 3 * gflop / r.us * 1e-3 * 1e3);
    // correctness of the last variant on a sample
    std::vector<float> y((size_t)M * N);
    CK(hipMemcpy(y.data(), dY, y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int s = 0; s < 2000; ++s) {
        const int r = (int)((uint64_t)rand() * 2654435761u % M), c = rand() % N;
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)a[(size_t)r * K + k] * (double)w[(size_t)c * K + k];
        worst = fmax(worst, fabs(ref - (double)y[(size_t)r * N + c]));
    }
    printf("  max |err| vs fp64 on 2000 samples: %.3e\n", worst);
    // the encode pass
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const dim3 grid((M + 63) / 64);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(enc_kernel, grid, dim3(256), 0, 0, dy2, dsc, dsh, dA2, M, Mp, K);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(enc_kernel, grid, dim3(256), 0, 0, dy2, dsc, dsh, dA2, M, Mp, K);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint32_t> back(ai.size());
        CK(hipMemcpy(back.data(), dA2, back.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int c = 0; c < K / 32; ++c)
            for (int pk = 0; pk < 8; ++pk)
                for (int r = 0; r < M; ++r)
                    for (int q = 0; q < 4; ++q) {
                        const size_t o = (((size_t)c * 8 + pk) * Mp + r) * 4 + q;
                        bad += back[o] != ai[o];
                    }
        printf("  encode pass (BN + ReLU + split, %.1f MB in, %.1f MB out): %8.2f us   image mismatches %zu\n", (double)M * K * 4e-6,
               (double)M * K * 4e-6, ms * 1000.f / it, bad);
    }
    return 0;
}
