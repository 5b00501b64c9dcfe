// Backward of one PointNet scale in entry space (autograd of models/det_base.py:75-101,134-157).
//
// Entry-space BatchNorm backward (derivation + CPU proof: tests/entry_ref.py, tests/test_entry_space_math.py):
//   dz[e] = G[e] * [z[e] > 0]                      G = gradient summed over the duplicates of entry e
//   dbeta = sum_e dz,  dgamma = sum_e dz * xhat
//   dy[e] = gamma*rstd * (dz[e] - w_e*dbeta/M - w_e*xhat[e]*dgamma/M)      (w_e = multiplicity, M = B*L*K)
//   dW    = sum_e dy[e] (x) a_prev[e],   G_prev[e] = W^T dy[e]
//
//   poolbwd        : routes dfeat to the max rows (gmax), dbeta3/dgamma3
//   dgrad<3>       : builds dy3 while staging (writes it once), G2 = dy3 . W3 on MFMA, ReLU mask from y2,
//                    writes dz2, dbeta2/dgamma2 in the epilogue
//   wgrad<3>       : dW3 = dy3^T . relu(bn2(y2)) (split over rows, partials + deterministic reduce)
//   dgrad<2>       : dy2 built while staging, G1 = dy2 . W2, ReLU mask recomputed from u, epilogue
//                    accumulates Q = sum dz1 * (1,u) -- enough for dW1/dgamma1/dbeta1 (conv1 is linear in u)
//   wgrad<2>       : dW2 = dy2^T . relu(bn1(conv1(u)))
//   l1_finalize    : dW1, dgamma1, dbeta1 from Q and the forward's input moments
#define FCN_TUNING_PNP
#include "gemm_tile.h"

#define LDT 129                 // transposed (k-major) staging of a 128-row tile
#define MAXC 512
#define WG_ROWS 128             // rows of one row tile (wgrad splits are multiples of it)
#define WG_TMAX 256             // row tiles per weight-gradient split (their lookup table lives in LDS)
#define PWB 16                  // windows per poolbwd workgroup (4 per wave, their loads issued together)
#ifndef FCN_WIDE_TILES
#define FCN_WIDE_TILES 0
#endif
#ifndef FCN_DG_WIDE
#define FCN_DG_WIDE 1         // 64 x 256 data-gradient tiles (8 waves) where the previous layer has 256 channels; 0: 64 x 128 everywhere
#endif
#ifndef FCN_WG_PITCH_PAD
#define FCN_WG_PITCH_PAD 0   // extra dwords per LDS row of the weight-gradient operand tiles (4 = the pitch of rounds 2-4 for 64*T wide tiles)
#endif
#ifndef FCN_WG3_OCC
#define FCN_WG3_OCC 3        // ... and the layer-3 one (stored dy3)
#endif
#ifndef FCN_WG2_OCC
#define FCN_WG2_OCC 2        // waves per SIMD the layer-2 (and rebuilt-dy3) weight-gradient kernels are compiled for
#endif
#ifndef FCN_DG2_OCC
#define FCN_DG2_OCC 3        // waves per SIMD the 64 x 128 data-gradient tile of layer 2 is compiled for (4: 128 VGPRs + 32 B scratch)
#endif

extern "C" int fcn_pn_wgrad_rows(void) { return WG_ROWS; }
extern "C" int fcn_stat_replicas(void) { return FCN_STAT_REP; }


// ------------------------------------------------------------------------------------------------
// One wave = one window x 64 channels (lane = channel); a workgroup covers PWB consecutive windows, stages
// its dfeat tile through LDS (dfeat is (B,C,L): 64-B runs along L), routes the gradient to the max rows and
// reduces dbeta3 / dgamma3 over its windows before one fp64 atomic pair per channel.
template <int S16>
__global__ __launch_bounds__(GT) void poolbwd_kernel(
    const float *__restrict__ dfeat, const int32_t *__restrict__ amax, const float *__restrict__ y3,
    const float *__restrict__ bn3, float *__restrict__ gmax, double *__restrict__ bstat, int rep_stride,
    int L, int cap, int C3, int CT, int nlc, int ywin)           // ywin: gmax holds the winners' pre-BN values (pool_keys_kernel)
{
    __shared__ float dS[64 * (PWB + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, l0 = blockIdx.x * PWB, c0 = blockIdx.y * 64;
    for (int f = tid; f < 64 * PWB && !nlc; f += GT) {
        const int cc = f / PWB, wl = f % PWB, l = l0 + wl;
        dS[cc * (PWB + 1) + wl] = (l < L) ? dfeat[((int64_t)b * CT + c0 + cc) * L + l] : 0.f;
    }
    __syncthreads();
    const int c = c0 + lane;
    const float mean = bn3[2 * C3 + c], rstd = bn3[3 * C3 + c];
    float sB = 0.f, sG = 0.f;
    // the wave's 4 windows: argmax rows first, then the gathers they address -- two dependent round trips in all
    constexpr int NW = PWB / 4;
    int am[NW];
    float gg[NW], yy[NW];
    // unconditional loads from clamped positions (a "load or default" select makes the compiler wait for each load at once):
    // the four arg-max loads, then the four gathers they address, are in flight together
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int l = l0 + wave + 4 * q;
        const int64_t o = ((int64_t)b * L + min(l, L - 1)) * C3 + c;
        am[q] = amax[o];
        gg[q] = nlc ? dfeat[o] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        if (l0 + wave + 4 * q >= L) am[q] = -1;
        if (ywin) yy[q] = gmax[((int64_t)b * L + min(l0 + wave + 4 * q, L - 1)) * C3 + c];      // (read here, overwritten below by the same lane)
        else yy[q] = lds1e<(S16 ? MM_BF16S : MM_F32)>(y3, ((int64_t)b * cap + max(am[q], 0)) * C3 + c);
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int wl = wave + 4 * q, l = l0 + wl;
        if (l >= L) continue;
        float g = 0.f;
        if (am[q] >= 0) {
            g = nlc ? gg[q] : dS[lane * (PWB + 1) + wl];
            const float xh = (yy[q] - mean) * rstd;
            sB += g;
            sG = fmaf(g, xh, sG);
        }
        gmax[((int64_t)b * L + l) * C3 + c] = g;
    }
    __syncthreads();
    float *red = dS;
    red[tid * 2] = sB;
    red[tid * 2 + 1] = sG;
    __syncthreads();
    if (tid < 64) {
        double a = 0.0, g = 0.0;
        for (int w = 0; w < 4; ++w) {
            a += (double)red[(w * 64 + tid) * 2];
            g += (double)red[(w * 64 + tid) * 2 + 1];
        }
        double *br = bstat + (int64_t)((blockIdx.x + blockIdx.z) % FCN_STAT_REP) * rep_stride;
        atomic_add_f64(&br[c0 + tid], a);
        atomic_add_f64(&br[C3 + c0 + tid], g);
    }
}

// ------------------------------------------------------------------------------------------------
struct DgradArgs {
    const float4 *ent;      // (B,cap)
    const int32_t *woff;    // (B,L+1)
    const int32_t *tiles;   // live-tile list
    const int32_t *ewin;    // (B,cap)            LAYER 3
    const float *ycur;      // y3 (LAYER 3) / y2 (LAYER 2): pre-BN output of the layer being differentiated
    const int32_t *amax;    // (B,L,C3)           LAYER 3
    const float *gmax;      // (B,L,C3)           LAYER 3
    const float *dzcur;     // dz2 (B,cap,C2)     LAYER 2
    FcnBnBwd cb;            // BN backward of the layer being differentiated (width CRED), finalised HERE by every workgroup
    const u32x4 *Wenc;      // data-gradient image of the conv weight (pn_pack_*, pointnet_fwd.hip): [CRED/32][2][4][CPREV]
    float *dybuf;           // LAYER 3: dy3 (B,cap,C3) written by the first column block
    const float *yprev;     // LAYER 3: y2 (B,cap,C2)
    const float *bn_prev;   // scale, shift, mean, rstd of the previous layer's BN (width CPREV)
    const float *W1;        // LAYER 2: (C1,3)
    float *dzprev;          // LAYER 3: dz2 out (B,cap,C2)
    double *bstat_prev;     // LAYER 3: dbeta2[C2], dgamma2[C2]; LAYER 2: Q[4][C1] -- replica 0 (stride cb.rep_stride)
    int L, cap, CRED, CPREV, tps;
};

// Tile = (64*MT) rows x (32*NT*WN) columns of the previous layer; 2 x WN waves, each MT x NT MFMA tiles.
// <L,1,2,2> (64 x 128) and <L,1,1,2> (64 x 64) stay under 168 VGPRs: 3 workgroups per CU instead of the 2 that the
// 128 x 128 tile's 220 VGPRs allow -- scale 4's 570 big tiles were 1.1 waves of 512 slots (two rounds, the second
// almost empty); 1140 half tiles on 768 slots are 1.5.
// Operands in the kb-major images of gemm_tile.h: dy rows built 8 reduction-adjacent channels at a time (one ds_write_b128
// per part), the weight from its pre-encoded data-gradient image (plain 16-byte copies), fragments by ds_read_b128.
// LDS bytes of a data-gradient workgroup: operand images + BN-backward coefficients + the tile rows' (ux,uy,uz,w)
template <int MT, int NT, int WN>
struct DgradLds {
    static constexpr int TM = 64 * MT, TN = 32 * NT * WN;
    static constexpr int U4 = KbTile<TM>::U4 + KbTile<TN>::U4;
    static constexpr int BYTES = U4 * 16 + 5 * MAXC * 4 + TM * 16;
};

// The body takes its workgroup index and LDS explicitly: dgrad_kernel below is the plain launch, pn_mid_kernel runs it as one
// ROLE beside the two weight-gradient GEMMs of the same scale (they all depend on the layer-3 data-gradient launch only).
// DZ3 (LAYER 3 only): the upstream gradient of layer 3 arrives DENSE, per entry row -- dzcur = dz3 (B,cap,C3), already masked by
// the ReLU -- instead of routed through the max-pool's arg-max / gradient maps: the backward of the reference's un-pooled module
// return (PointNetModule.forward, models/det_base.py:62-103; fcn_pn_backward_dense).
template <int MM, int LAYER, int MT, int NT, int WN, int DZ3 = 0>
__device__ __forceinline__ void dgrad_body(const DgradArgs &a, const int bid, unsigned char *smem_)
{
    static_assert(!DZ3 || LAYER == 3, "DZ3 is a form of the layer-3 data gradient");
    constexpr bool MAPS = LAYER == 3 && !DZ3;     // dz of layer 3 from the pooled maps
    constexpr int NTHR = 128 * WN;
    constexpr int TM = 64 * MT;               // rows of the tile
    constexpr int TN = 32 * NT * WN;
    constexpr int LDRA = KbTile<TM>::LDR, LDRB = KbTile<TN>::LDR;
    constexpr int NA4 = TM * 8 / NTHR;        // 16-byte pieces of the A tile per thread per chunk
    constexpr int NB = TN * 8 / NTHR;         // u32x4 of the encoded weight per thread per chunk
    constexpr int SUB = 128 / TM;             // workgroups per 128-row tile of the live-tile list
    constexpr int LDSU4 = KbTile<TM>::U4 + KbTile<TN>::U4;
    u32x4 *lds4 = (u32x4 *)smem_;
    float *coefS = (float *)(smem_ + LDSU4 * 16);
    float4 *uS = (float4 *)(smem_ + LDSU4 * 16 + 5 * MAXC * 4);      // (ux,uy,uz,w) of the tile rows
    u32x4 *Ab = lds4, *Bb = lds4 + KbTile<TM>::U4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ny = a.CPREV / TN;                  // XCD order, column tiles fastest (see fcn_xcd_tile)
    if (bid == 0 && a.cb.dgamma) {                // workgroup 0 always exists (the grid is padded): it exports dgamma / dbeta
        for (int c = tid; c < a.CRED; c += NTHR) {
            a.cb.dgamma[c] = (float)fcn_rep_sum(a.cb.bstat + a.CRED + c, a.cb.rep_stride);
            a.cb.dbeta[c] = (float)fcn_rep_sum(a.cb.bstat + c, a.cb.rep_stride);
        }
    }
    const int xt = fcn_xcd_tile(bid, SUB * a.tiles[0] * ny);
    if (xt < 0) return;
    const int bxi = xt / ny, byi = xt % ny;
    const int lt = bxi / SUB, sub = bxi % SUB;
    const int code = a.tiles[4 + lt];
    const int b = code / a.tps, t = code % a.tps;
    const int nent = a.woff[(int64_t)b * (a.L + 1) + a.L];
    const int row0 = t * 128 + sub * TM;
    const int nvalid = min(TM, nent - row0);
    if (nvalid <= 0) return;
    const int64_t grow0 = (int64_t)b * a.cap + row0;
    const int k0 = byi * TN;                      // first output column (channel of the previous layer)
    const int CRED = a.CRED, CPREV = a.CPREV;

    PNP_DECL;
    for (int c = tid; c < CRED; c += NTHR) {
        float cf[5];
        fcn_bnbwd_coef(a.cb, CRED, c, cf, false);
#pragma unroll
        for (int q = 0; q < 5; ++q) coefS[q * CRED + c] = cf[q];
    }
    if (tid < TM) uS[tid] = (tid < nvalid) ? a.ent[grow0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    // loads are UNCONDITIONAL on a clamped row (a "load or zero" branch makes hipcc wait for the load at once); rows
    // past nvalid are zeroed when the registers go to LDS
    // 32-bit element offsets (launch_dgrad checks B * cap * max(CRED, CPREV) < 2^31): half the registers and address
    // arithmetic of int64
    // Addresses = a wave-uniform base (the tile's first row / the frustum's first window, advanced by the chunk: scalar arithmetic)
    // + a loop-invariant 32-bit byte offset per lane: no vector address arithmetic in the loop (gemm_tile.h lds4b)
    unsigned arowB[NA4];  // byte offset of this thread's piece (clamped row) from the tile's first row in a (rows, CRED) buffer
    unsigned wbaseB[NA4]; // LAYER 3: byte offset of the piece in its row's window from the frustum's first window of the (B, L, CRED) maps
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int f = tid + NTHR * i;
        const int rc = min(f >> 3, nvalid - 1);
        arowB[i] = (unsigned)(rc * CRED + 4 * (f & 7)) * St<MM>::bytes;
        wbaseB[i] = 0;
        if constexpr (MAPS) wbaseB[i] = (unsigned)(a.ewin[grow0 + rc] * CRED + 4 * (f & 7)) * 4u;
    }
    const int64_t tile0 = grow0 * CRED;                     // elements in front of the tile's first row
    const int64_t map0 = (int64_t)b * a.L * CRED;           // elements in front of the frustum's first window
    __syncthreads();

    f32x16 acc[MT][NT];
    acc_zero<MT, NT>(acc);
    v4f ry[NA4], rz[NA4];
    v4i rm[NA4];
    u32x4 rw[NB];
    const int nchunk = CRED / KC;
    // weight image item f = tid + NTHR*i: column f % TN, (plane, k-block) f / TN
    unsigned woffB[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) woffB[i] = (unsigned)((tid % TN) + (tid / TN + i * (NTHR / TN)) * CPREV) * 16u;
    const u32x4 *wsrc = a.Wenc + k0;

#define DGRAD_LOAD(cc)                                                                                                \
    {                                                                                                                 \
        const int nq_ = (cc) * KC;                                                                                    \
        if (!((FCN_XB & 1) && (cc) > 0))                                                                              \
        {                                                                                                             \
            const float *yb_ = st_ptr<MM>(a.ycur, tile0 + nq_);                                                       \
            const int *mb_ = MAPS ? a.amax + map0 + nq_ : nullptr;                                                    \
            const float *gb_ = MAPS ? a.gmax + map0 + nq_ : st_ptr<MM>(a.dzcur, tile0 + nq_);                         \
            _Pragma("unroll") for (int i = 0; i < NA4; ++i) {                                                         \
                ry[i] = lds4b<MM>(yb_, arowB[i]);                                                                     \
                if constexpr (MAPS) {                                                                                 \
                    rm[i] = *(gv4ip)((const char *)mb_ + fcn_opaque_v32(wbaseB[i]));                                  \
                    rz[i] = *(gv4fp)((const char *)gb_ + fcn_opaque_v32(wbaseB[i]));                                  \
                } else {                                                                                              \
                    rz[i] = lds4b<MM>(gb_, arowB[i]);                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
        if (!((FCN_XB & 2) && (cc) > 0))                                                                              \
        {                                                                                                             \
            const char *wb_ = (const char *)(wsrc + (int64_t)(cc) * 8 * CPREV);                                       \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) rw[i] = *(gu4p)(wb_ + fcn_opaque_v32(woffB[i]));           \
        }                                                                                                             \
    }

    PNP_ADD(0);                                   // 0: prologue (tables, coefficients, first barrier)
    DGRAD_LOAD(0);
    for (int c = 0; c < nchunk; ++c) {
        PNP_ADD(1);                               // 1: issue of the global loads (+ loop overhead)
        if ((FCN_XB & 4) && c > 0) goto staged;
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            const int f = tid + NTHR * i;
            const int r = f >> 3, kq = f & 7;
            const bool ok = r < nvalid;
            const float w = uS[r].w;
            const int rloc = row0 + r;
            const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
            const float zv[4] = {rz[i].x, rz[i].y, rz[i].z, rz[i].w};
            int mv[4] = {0, 0, 0, 0};
            if constexpr (MAPS) { mv[0] = rm[i].x; mv[1] = rm[i].y; mv[2] = rm[i].z; mv[3] = rm[i].w; }
            const int nb = c * KC + 4 * kq;
            float cfv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f c4 = *(const v4f *)(coefS + q * CRED + nb);
                cfv[q][0] = c4.x; cfv[q][1] = c4.y; cfv[q][2] = c4.z; cfv[q][3] = c4.w;
            }
            float dv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dz = zv[j];
                if constexpr (MAPS) dz = (mv[j] == rloc) ? zv[j] : 0.f;
                // (rows past nvalid hold a live row's values again -- the loads are clamped -- and are not zeroed: a row of this
                // operand reaches only its own output row, which the epilogue neither stores nor counts)
                dv[j] = fcn_bn_dy(cfv[0][j], cfv[1][j], cfv[2][j], cfv[3][j], dz, yv[j], w);
            }
            kb_store4<MM_ENC_A, LDRA>(Ab, r, kq, dv[0], dv[1], dv[2], dv[3]);
            if constexpr (LAYER == 3) {
                if (ok && byi == 0 && a.dybuf && !(FCN_XB & 128)) {
                    const v4f d0 = {dv[0], dv[1], dv[2], dv[3]};
                    sts4e<MM>(a.dybuf, (grow0 + r) * CRED + nb, d0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + NTHR * i;
            Bb[(f / TN) * LDRB + (f % TN)] = rw[i];
        }
    staged:
        PNP_ADD(2);                               // 2: wait for the loads + operand transform + LDS stores
        __syncthreads();
        PNP_ADD(3);                               // 3: barrier in front of the MFMA phase
        if (c + 1 < nchunk) DGRAD_LOAD(c + 1);
        PNP_ADD(1);
        if (!(FCN_XB & 8)) mma_chunk_kb<MM, MT, NT, LDRA, LDRB>(Ab, Bb, wm * 32 * MT, wn * 32 * NT, acc);
        PNP_ADD(4);                               // 4: LDS operand reads + MFMAs
        __syncthreads();
        PNP_ADD(3);                               // (3: both barriers)
    }
#undef DGRAD_LOAD

    PNP_ADD(3);
    if (FCN_XB & 16) { if (acc[0][0][0] == 123.456f) a.bstat_prev[0] = 0.0; return; }
    // ---- epilogue: ReLU mask of the previous layer, its BN-backward statistics
    constexpr int NS = (LAYER == 3) ? 2 : 4;
    float st[NT][NS];
    // (LAYER 3 through a transposition patch -- the previous layer's output in and the masked gradient out as 16-byte accesses,
    // 4 columns per lane -- measured 20 % (64 x 128 tiles) to 100 % (64 x 64) SLOWER than this register form on MI355X: four
    // loads in flight per lane instead of sixteen, and 24 cross-lane sums per column tile instead of 2)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = k0 + wn * 32 * NT + nt * 32 + l31;
        const float ps = a.bn_prev[col], pt = a.bn_prev[CPREV + col];
#pragma unroll
        for (int q = 0; q < NS; ++q) st[nt][q] = 0.f;
        if constexpr (LAYER == 3) {
            const float pm = a.bn_prev[2 * CPREV + col], pr = a.bn_prev[3 * CPREV + col];
            // ALL loads of the previous layer's output first, then the masked stores: interleaved, every store to dzprev may
            // alias the next yprev load as far as the compiler knows, and the 32 load -> store pairs of a lane ran one memory
            // round trip after the other -- 37-50 % of a workgroup's cycles (tools/pn_probe.py) for 16 KB in and out
            float yv[MT][16];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = min(wm * 32 * MT + mt * 32 + acc_row(reg, lh), nvalid - 1);      // clamped, unconditional
                    yv[mt][reg] = lds1e<MM>(a.yprev, (grow0 + row) * CPREV + col);
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = wm * 32 * MT + mt * 32 + acc_row(reg, lh);
                    if (row < nvalid) {
                        const float y = yv[mt][reg];
                        const float dz = st_round<MM>((fmaf(ps, y, pt) > 0.f) ? acc[mt][nt][reg] : 0.f);       // as stored
                        if (!(FCN_XB & 32)) sts1e<MM>(a.dzprev, (grow0 + row) * CPREV + col, dz);
                        st[nt][0] += dz;
                        st[nt][1] = fmaf(dz, (y - pm) * pr, st[nt][1]);
                    }
                }
        } else {
            const float al[3] = {ps * a.W1[3 * col], ps * a.W1[3 * col + 1], ps * a.W1[3 * col + 2]};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = wm * 32 * MT + mt * 32 + acc_row(reg, lh);
                    if (row < nvalid) {
                        const float4 u = uS[row];
                        const float dz = (l1_pre(al, pt, u.x, u.y, u.z) > 0.f) ? acc[mt][nt][reg] : 0.f;
                        st[nt][0] += dz;
                        st[nt][1] = fmaf(dz, u.x, st[nt][1]);
                        st[nt][2] = fmaf(dz, u.y, st[nt][2]);
                        st[nt][3] = fmaf(dz, u.z, st[nt][3]);
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) st[nt][q] += __shfl_xor(st[nt][q], 32, 64);
    }
    PNP_ADD(5);                                   // 5: epilogue -- ReLU mask (loads of the previous layer's output), dz stores
    float *red = (float *)lds4;      // [wn][nt][l31][NS], written by wm == 1 (the operand images are free after the last barrier)
    if (wm == 1 && lh == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < NS; ++q) red[(((wn * NT + nt) * 32 + l31) * NS) + q] = st[nt][q];
    }
    __syncthreads();
    if (wm == 0 && lh == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = k0 + wn * 32 * NT + nt * 32 + l31;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const double v = (double)st[nt][q] + (double)red[(((wn * NT + nt) * 32 + l31) * NS) + q];
                double *br = a.bstat_prev + (int64_t)fcn_rep_id() * a.cb.rep_stride;
                if (!(FCN_XB & 64)) atomic_add_f64(&br[q * CPREV + col], v);
                else if (v == 123.456) br[0] = v;
            }
        }
    }
    PNP_FLUSH(((unsigned long long)LAYER << 48) | ((unsigned long long)CRED << 32) | ((unsigned long long)CPREV << 16) |
              (unsigned long long)nvalid);
}

template <int MM, int LAYER, int MT, int NT, int WN, int DZ3 = 0>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(MT * NT <= 2 ? (LAYER == 2 ? FCN_DG2_OCC : 3) : 2, 4)))
void dgrad_kernel(DgradArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[DgradLds<MT, NT, WN>::BYTES];
    dgrad_body<MM, LAYER, MT, NT, WN, DZ3>(a, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float4 *ent;
    const int32_t *woff;
    const int32_t *tiles;   // live-tile list
    const float *dy;        // LAYER 3: dy3 (B,cap,C3) as the data-gradient GEMM stored it, or nullptr: REBUILT here (RC)
    const float *dz;        // LAYER 2: dz2 (B,cap,C2)
    const float *ycur;      // LAYER 2: y2 (for xhat2); LAYER 3 RC: y3
    FcnBnBwd cb;            // LAYER 2: BN2 backward (LAYER 3 RC: BN3 backward), finalised by every workgroup
    const int32_t *ewin;    // LAYER 3 RC: window of each row, and the arg-max / routed-gradient maps (B,L,C3) of the max-pool
    const int32_t *amax;
    const float *gmax;
    const float *yprev;     // LAYER 3: y2 -> a2
    const float *bn_prev;   // scale, shift of the previous layer's BN
    const float *W1;        // LAYER 2
    float *partial;         // (nsplit, COUT, CIN)
    int L, cap, COUT, CIN, tps;
};

// four stored elements (fp32, or bf16 in the bf16 throughput mode) as a float4
template <int MM>
__device__ __forceinline__ float4 ld4f(const float *base, int64_t e)
{
    const v4f v = lds4e<MM>(base, e);
    return make_float4(v.x, v.y, v.z, v.w);
}

// ... the same through a wave-uniform base pointer + a 32-bit per-lane byte offset (gemm_tile.h lds4b)
template <int MM>
__device__ __forceinline__ float4 ld4fb(const float *sbase, unsigned boff)
{
    const v4f v = lds4b<MM>(sbase, boff);
    return make_float4(v.x, v.y, v.z, v.w);
}

// dW[n][k] = sum_rows dy[row][n] * a_prev[row][k]; workgroup tile (64*MT) x (64*NT).  Split s reduces the
// rows of live tiles [s*tpb, (s+1)*tpb) and writes one partial; wgrad_reduce sums the live partials in a fixed
// order (deterministic, no float atomics).
// RC (LAYER 3): dy3 is not read from memory but REBUILT while staging, the way the data-gradient GEMM builds it -- from y3, the
// row's multiplicity, the max-pool's arg-max / routed-gradient maps at the row's window and the BN3-backward coefficients -- so
// that nothing has to write (B, cap, C3) floats for this kernel to read back: the same fp32 expression, bit-identical dW3.
// The window ids of a chunk's rows are fetched one chunk ahead (the map addresses depend on them).
template <int MT, int NT, int RC>
struct WgradLds {
    // Row pitch of the k-major operand tiles: a MULTIPLE OF 64 dwords -- every row of the 32-deep chunk is then within the 8-bit,
    // 64-dword-unit offsets of ds_read2st64_b32 from ONE base register per 32-column block (with the 64*MT + 4 pitch of rounds 2-4
    // two rows were the most a ds_read2_b32 reached: 33 v_add_u32 per chunk rebuilt bases in a loop that is vector-issue-bound)
    static constexpr int pitch(int T) { return (64 * T + 63) / 64 * 64 + (FCN_WG_PITCH_PAD); }
    static constexpr int LDA = pitch(MT), LDB = pitch(NT);
    static constexpr int BYTES = KC * (LDA + LDB) * 4 + (2 + (RC ? 2 : 0)) * WG_TMAX * 4;
};

template <int MM, int LAYER, int MT, int NT, int RC = 0>
__device__ __forceinline__ void wgrad_body(const WgradArgs &a, const int bx_, const int by_, const int bz_, const int gx_,
                                           unsigned char *smem_)
{
    constexpr bool XF = LAYER == 2 || RC;               // the A operand is transformed by a BatchNorm backward while staging
    constexpr int LDA = WgradLds<MT, NT, RC>::LDA, LDB = WgradLds<MT, NT, RC>::LDB;
    float *As = (float *)smem_;
    float *Bs = As + KC * LDA;
    // (first row, live rows) of the split's row tiles, looked up ONCE: per chunk, the walk tile list -> frustum -> live-row
    // count was two dependent memory round trips in front of every chunk's loads AND again in front of its staging, and the
    // "load or zero" branches behind it made the compiler wait for every load at once -- tools/pn_probe.py: 45-60 % of the
    // kernel's cycles between them, 13-19 % in the MFMA phase
    int *tG0 = (int *)(Bs + KC * LDB), *tLeft = tG0 + WG_TMAX;
    int *tBL = tLeft + WG_TMAX, *tR0 = tBL + (RC ? WG_TMAX : 0);      // RC: b * L and the tile's first row within its frustum

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = a.tiles[0];
    const int tpb = (ntile + gx_ - 1) / gx_;            // live tiles per split, balanced on the device
    const int t_beg = bx_ * tpb;
    if (t_beg >= ntile) return;
    const int t_end = min(ntile, t_beg + tpb);
    const int nq = (t_end - t_beg) * 4;                 // 32-row chunks to reduce
    const int n0 = by_ * 64 * MT, k0 = bz_ * 64 * NT;
    const int COUT = a.COUT, CIN = a.CIN;
    for (int i = tid; i < t_end - t_beg; i += GT) {     // (launch_wgrad keeps a split within WG_TMAX tiles)
        const int code = a.tiles[4 + t_beg + i];
        const int b = code / a.tps, t = code % a.tps;
        tG0[i] = b * a.cap + t * 128;
        tLeft[i] = a.woff[(int64_t)b * (a.L + 1) + a.L] - t * 128;
        if constexpr (RC) { tBL[i] = b * a.L; tR0[i] = t * 128; }
    }
    __syncthreads();

    // per-thread constant columns of the two staged operands
    const int acq = tid % (16 * MT), arr = tid / (16 * MT);
    const int bcq = tid % (16 * NT), brr = tid / (16 * NT);
    // a thread's float4s sit in reduction-adjacent row PAIRS (rows 2*arr + {0,1} + 2*ART*p): what enc2 packs together
    constexpr int ART = 16 / MT, BRT = 16 / NT;                  // threads along the rows of the A / B tile
#define WG_AROW(i) (2 * arr + ((i) & 1) + 2 * ART * ((i) >> 1))
#define WG_BROW(i) (2 * brr + ((i) & 1) + 2 * BRT * ((i) >> 1))
    float cf[5][4];
    float bs[4], bt[4], bal[4][3];
    if constexpr (XF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float c5[5];
            fcn_bnbwd_coef(a.cb, COUT, n0 + 4 * acq + j, c5, false);
#pragma unroll
            for (int q = 0; q < 5; ++q) cf[q][j] = c5[q];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * bcq + j;
        bs[j] = a.bn_prev[k];
        bt[j] = a.bn_prev[CIN + k];
        if constexpr (LAYER == 2) {
            bal[j][0] = bs[j] * a.W1[3 * k];
            bal[j][1] = bs[j] * a.W1[3 * k + 1];
            bal[j][2] = bs[j] * a.W1[3 * k + 2];
        }
    }

    PNP_DECL;
    f32x16 acc[MT][NT];
    acc_zero<MT, NT>(acc);
    float4 ra[2 * MT], ra2[2 * MT], rb4[2 * NT];
    float rwt[2 * MT];
    v4i rm[RC ? 2 * MT : 1];                   // RC: arg-max rows of the row's window at this thread's four channels
    int rwin[RC ? 2 * MT : 1], rwin_n[RC ? 2 * MT : 1];      // RC: windows of the chunk being loaded / of the next one

    // chunk q -> (global row of its first row, number of valid rows left in its tile from there)
    // (32-bit element offsets throughout: launch_wgrad checks B * cap * max(COUT, CIN) < 2^31 -- the 64-bit multiplies of
    // the per-chunk address arithmetic were a visible part of the load-issue phase)
    auto chunk_rows = [&](int q, int &g0, int &left) __attribute__((always_inline)) {
        const int r0 = (q & 3) * KC;
        g0 = tG0[q >> 2] + r0;
        left = tLeft[q >> 2] - r0;                             // may be <= 0 for the tail chunks of a tile
    };

    // RC: window ids of chunk q's rows -> rwin_n (same clamping as the data loads)
    auto load_win = [&](int q) __attribute__((always_inline)) {
        if constexpr (RC) {
            int g0, left;
            chunk_rows(q, g0, left);
            const int lastr = max(left, 1) - 1;
            if (left <= 0) g0 = tG0[q >> 2];
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) rwin_n[i] = a.ewin[g0 + min(WG_AROW(i), lastr)];
        }
    };

    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        int g0, left;
        chunk_rows(q, g0, left);
        // unconditional loads from a clamped row; rows past `left` are zeroed when the registers go to LDS.  A chunk without
        // a live row (the tail of a tile: cap need not be a multiple of 128, so its rows may lie past the buffer) reads the
        // tile's first row, which is live.
        const int lastr = max(left, 1) - 1;
        if (left <= 0) g0 = tG0[q >> 2];
        int bl = 0;
        if constexpr (RC) {
            bl = tBL[q >> 2];
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) rwin[i] = rwin_n[i];        // (requested one chunk ago)
        }
        // the chunk's first row is wave-uniform: its part of every address goes into SCALAR base pointers, the lanes add a 32-bit
        // byte offset inside the chunk (no 64-bit vector arithmetic per load)
        constexpr unsigned SB = St<MM>::bytes;
        const int g0s = __builtin_amdgcn_readfirstlane(g0);
        const float *abase = st_ptr<MM>(LAYER == 3 ? a.dy : a.dz, (int64_t)g0s * COUT);
        const float *ybase = st_ptr<MM>(a.ycur, (int64_t)g0s * COUT);
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int rr = min(WG_AROW(i), lastr);
            const int o = RC ? (int)fcn_mad24((unsigned)(g0 + rr), (unsigned)COUT, (unsigned)(n0 + 4 * acq)) : 0;      // (rows < 2^24: launch_wgrad)
            const unsigned ob = fcn_mad24((unsigned)rr, (unsigned)COUT * SB, (unsigned)(n0 + 4 * acq) * SB);
            if constexpr (RC) {
                const int om = (int)fcn_mad24((unsigned)(bl + rwin[i]), (unsigned)COUT, (unsigned)(n0 + 4 * acq));
                ra2[i] = ld4f<MM>(a.ycur, o);
                rm[i] = ldg4i(a.amax + om);
                const v4f g4 = ldg4(a.gmax + om);
                ra[i] = make_float4(g4.x, g4.y, g4.z, g4.w);
                rwt[i] = a.ent[g0 + rr].w;
            } else if constexpr (LAYER == 3) {
                ra[i] = ld4fb<MM>(abase, ob);
            } else {
                ra[i] = ld4fb<MM>(abase, ob);
                ra2[i] = ld4fb<MM>(ybase, ob);
                rwt[i] = a.ent[g0 + rr].w;
            }
        }
        if constexpr (RC) {
            if (q + 1 < nq) load_win(q + 1);
        }
        const float *pbase = LAYER == 3 ? st_ptr<MM>(a.yprev, (int64_t)g0s * CIN) : nullptr;
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const int rr = min(WG_BROW(i), lastr);
            if constexpr (LAYER == 3) rb4[i] = ld4fb<MM>(pbase, fcn_mad24((unsigned)rr, (unsigned)CIN * SB, (unsigned)(k0 + 4 * bcq) * SB));
            else rb4[i] = a.ent[g0 + rr];
        }
    };

    PNP_ADD(0);                                   // 0: prologue
    load_win(0);
    load_chunk(0);
    for (int q = 0; q < nq; ++q) {
        PNP_ADD(1);                               // 1: chunk lookup + issue of the global loads
        int g0_, left;
        chunk_rows(q, g0_, left);
        PNP_ADD(5);                               // 5: chunk lookup at the loop top (tile list -> live rows)
        v4f sa[2 * MT], sb[2 * NT];
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int rr = WG_AROW(i);
            const bool ok = rr < left;
            const v4f rv = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            v4f v = ok ? rv : zero4();
            if constexpr (XF) {
                float dzv[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                if constexpr (RC) {          // the routed gradient counts at the window's arg-max row only
                    const int rloc = tR0[q >> 2] + (q & 3) * KC + rr;
                    const int mv[4] = {rm[i].x, rm[i].y, rm[i].z, rm[i].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) dzv[j] = (mv[j] == rloc) ? dzv[j] : 0.f;
                }
                const float yv[4] = {ra2[i].x, ra2[i].y, ra2[i].z, ra2[i].w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = fcn_bn_dy(cf[0][j], cf[1][j], cf[2][j], cf[3][j], dzv[j], yv[j], rwt[i]);
                }
                v4f ov = {o[0], o[1], o[2], o[3]};
                v = ok ? ov : zero4();
            }
            sa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2 * MT; i += 2) {
            v4f hi, lo;
            enc2x4<MM_ENC_A>(sa[i], sa[i + 1], hi, lo);               // rows (2p, 2p + 1) = pair p: gemm_tile.h "pair-plane" order
            sts4(As + mma_row_hi<KC>(WG_AROW(i) >> 1) * LDA + 4 * acq, hi);
            sts4(As + mma_row_lo<KC>(WG_AROW(i) >> 1) * LDA + 4 * acq, lo);
        }
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const int rr = WG_BROW(i);
            const float kp = fcn_keep(rr < left);         // ReLU + row mask as one v_med3_f32 per element
            float o[4];
            if constexpr (LAYER == 3) {
                const float yv[4] = {rb4[i].x, rb4[i].y, rb4[i].z, rb4[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fcn_relu_keep(fmaf(bs[j], yv[j], bt[j]), kp);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = fcn_relu_keep(l1_pre(bal[j], bt[j], rb4[i].x, rb4[i].y, rb4[i].z), kp);
            }
            v4f ov = {o[0], o[1], o[2], o[3]};
            sb[i] = ov;
        }
#pragma unroll
        for (int i = 0; i < 2 * NT; i += 2) {
            v4f hi, lo;
            enc2x4<MM_ENC_A>(sb[i], sb[i + 1], hi, lo);
            sts4(Bs + mma_row_hi<KC>(WG_BROW(i) >> 1) * LDB + 4 * bcq, hi);
            sts4(Bs + mma_row_lo<KC>(WG_BROW(i) >> 1) * LDB + 4 * bcq, lo);
        }
        PNP_ADD(2);                               // 2: wait for the loads + operand transform + LDS stores
        __syncthreads();
        PNP_ADD(3);                               // 3: barriers
        if (q + 1 < nq) load_chunk(q + 1);
        PNP_ADD(1);
        mma_chunk<MM, MT, NT, LDA, LDB>(As, Bs, wm * 32 * MT, wn * 32 * NT, acc);
        PNP_ADD(4);                               // 4: LDS operand reads + MFMAs
        __syncthreads();
        PNP_ADD(3);
    }
#undef WG_AROW
#undef WG_BROW

    float *out = a.partial + (int64_t)bx_ * COUT * CIN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = n0 + wm * 32 * MT + mt * 32 + acc_row(reg, lh);
                const int k = k0 + wn * 32 * NT + nt * 32 + l31;
                out[(int64_t)n * CIN + k] = acc[mt][nt][reg];
            }
    PNP_FLUSH(((unsigned long long)(10 + LAYER) << 48) | ((unsigned long long)COUT << 32) | ((unsigned long long)CIN << 16) |
              (unsigned long long)64);
}

template <int MM, int LAYER, int MT, int NT, int RC = 0>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu((LAYER == 3 && !RC) ? FCN_WG3_OCC : FCN_WG2_OCC, 4))) void wgrad_kernel(WgradArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WgradLds<MT, NT, RC>::BYTES];
    wgrad_body<MM, LAYER, MT, NT, RC>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, smem);
}

// ------------------------------------------------------------------------------------------------
// The three GEMMs of a scale's backward that depend on the layer-3 data-gradient launch ONLY -- conv3's weight gradient (reads
// dy3), conv2's data gradient and conv2's weight gradient (both read dz2 and the BN2-backward sums) -- as ROLES of one launch:
// workgroups [0, n_g2) run dgrad<2> tiles, [n_g2, n_g2 + n_w3) the splits of wgrad<3>, the rest those of wgrad<2>.  On one stream
// the three used to queue behind each other with their reduces in between (8 dependent launches per scale; the narrow scales'
// chains are a string of 5-30 us kernels that ends the step's backward); here the chain is poolbwd -> dgrad<3> -> this launch ->
// reduce, reduce, finalise.  The roles share the LDS (the largest of the three) and the register budget of the widest.
struct PnMidArgs {
    DgradArgs g2;
    WgradArgs w3, w2;
    int n_g2, n_w3;            // workgroups of the first two roles (n_g2 a multiple of 8: the XCD-aware tile order of the role)
    int ns3, oy3, ns2, oy2;    // split counts and row-tile counts of the two weight-gradient grids (x = split, y, z)
};
template <int A, int B>
struct CMax { static constexpr int V = A > B ? A : B; };

template <int MM, int DNT, int W3M, int W3N, int W2M, int W2N>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 4))) void pn_mid_kernel(PnMidArgs m)
{
    constexpr int LB = CMax<CMax<DgradLds<1, DNT, 2>::BYTES, WgradLds<W3M, W3N, 0>::BYTES>::V, WgradLds<W2M, W2N, 0>::BYTES>::V;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LB];
    const int bid = (int)blockIdx.x;
    if (bid < m.n_g2) {
        dgrad_body<MM, 2, 1, DNT, 2>(m.g2, bid, smem);
    } else if (bid < m.n_g2 + m.n_w3) {
        const int i = bid - m.n_g2, bx = i % m.ns3, r = i / m.ns3;
        wgrad_body<MM, 3, W3M, W3N, 0>(m.w3, bx, r % m.oy3, r / m.oy3, m.ns3, smem);
    } else {
        const int i = bid - m.n_g2 - m.n_w3, bx = i % m.ns2, r = i / m.ns2;
        wgrad_body<MM, 2, W2M, W2N, 0>(m.w2, bx, r % m.oy2, r / m.oy2, m.ns2, smem);
    }
}

// out[i] = sum over the live splits in a fixed order (deterministic).  The 256-thread workgroup covers 256 / gr
// consecutive elements with gr split groups (gr = 1, 2, 4, 8 or 16 picked on the host from the split count): a small
// layer has few elements but hundreds of splits (a few independent loads per thread, group sum through LDS), a big one
// the opposite (one thread per element).
#define WR_T 256
// (each thread owns FOUR consecutive elements: 16-byte loads -- a quarter of the load instructions of the dword version and
// four times the bytes in flight per thread; 100 us of kernel time per step went through here at ~1.6 TB/s)
struct ReduceJob {
    const float *partial;
    const int32_t *tiles;
    float *out;
    int64_t nelem;
    int nsplit, gr, nblk;          // nblk: workgroups of this job
};
__device__ __forceinline__ void wgrad_reduce_body(const ReduceJob &j, const int rid, float *sh)
{
    const int gr = j.gr, per = WR_T / gr;
    const int x = threadIdx.x % per, y = threadIdx.x / per;
    const int64_t i = ((int64_t)rid * per + x) * 4;      // nelem is a multiple of 1024
    const int ntile = j.tiles[0];
    const int tpb = (ntile + j.nsplit - 1) / j.nsplit;
    const int nsp = tpb > 0 ? (ntile + tpb - 1) / tpb : 0;
    const float *p = j.partial + i;
    const int64_t nelem = j.nelem;
    v4f s0 = zero4(), s1 = zero4(), s2 = zero4(), s3 = zero4();
    int sp = y;
    for (; sp + 3 * gr < nsp; sp += 4 * gr) {
        s0 += ldg4(p + (int64_t)sp * nelem);
        s1 += ldg4(p + (int64_t)(sp + gr) * nelem);
        s2 += ldg4(p + (int64_t)(sp + 2 * gr) * nelem);
        s3 += ldg4(p + (int64_t)(sp + 3 * gr) * nelem);
    }
    for (; sp < nsp; sp += gr) s0 += ldg4(p + (int64_t)sp * nelem);
    v4f t = (s0 + s1) + (s2 + s3);
    if (gr > 1) {
        sts4(sh + 4 * (y * per + x), t);
        __syncthreads();
        if (y != 0) return;
        t = zero4();
        for (int q = 0; q < gr; ++q) t += *(const v4f *)(sh + 4 * (q * per + x));
    }
    sts4(j.out + i, t);
}
__global__ __launch_bounds__(WR_T) void wgrad_reduce_kernel(ReduceJob j)
{
    __shared__ __attribute__((aligned(16))) float sh[WR_T * 4];
    wgrad_reduce_body(j, (int)blockIdx.x, sh);
}

// dW1, dgamma1, dbeta1 from Q = sum_e dz1 (1, u) and the forward's weighted moments of u.
struct L1Args {
    const double *Qr;
    int rep_stride;
    const double *mom;
    const float *W1, *gamma, *bn1;
    int C;
    double M;
    float *dW1, *dgamma, *dbeta;
};
__device__ __forceinline__ void l1_finalize_body(const L1Args &a, const int c)
{
    const int C = a.C;
    if (c >= C) return;
    const double *Qr = a.Qr, *mom = a.mom;
    const int rep_stride = a.rep_stride;
    const double q0 = fcn_rep_sum(Qr + c, rep_stride);
    const double qu[3] = {fcn_rep_sum(Qr + C + c, rep_stride), fcn_rep_sum(Qr + 2 * C + c, rep_stride), fcn_rep_sum(Qr + 3 * C + c, rep_stride)};
    const double w[3] = {a.W1[3 * c], a.W1[3 * c + 1], a.W1[3 * c + 2]};
    const double mean = a.bn1[2 * C + c], rstd = a.bn1[3 * C + c];
    const double db = q0;
    const double dg = rstd * (w[0] * qu[0] + w[1] * qu[1] + w[2] * qu[2] - mean * q0);
    const double iM = 1.0 / a.M;          // (one fp64 division instead of twelve: the last launch of the scale's backward chain)
    const double mu[3] = {mom[1] * iM, mom[2] * iM, mom[3] * iM};
    const double m2[3][3] = {{mom[4] * iM, mom[5] * iM, mom[6] * iM},
                             {mom[5] * iM, mom[7] * iM, mom[8] * iM},
                             {mom[6] * iM, mom[8] * iM, mom[9] * iM}};
    const double kk = (double)a.gamma[c] * rstd;
    for (int j = 0; j < 3; ++j) {
        const double wm2 = w[0] * m2[0][j] + w[1] * m2[1][j] + w[2] * m2[2][j];
        // sum_e w x^ u_j = rstd * M * (W1_c . m2[:,j] - mean * mu_j);  sum_e w u_j = M mu_j
        const double v = kk * (qu[j] - db * mu[j] - dg * rstd * (wm2 - mean * mu[j]));
        a.dW1[3 * c + j] = (float)v;
    }
    a.dgamma[c] = (float)dg;
    a.dbeta[c] = (float)db;
}
__global__ void l1_finalize_kernel(L1Args a) { l1_finalize_body(a, (int)(blockIdx.x * blockDim.x + threadIdx.x)); }

// The TAIL of a scale's backward in one launch: the fixed-order sums of both weight gradients' split partials and (optionally) the
// layer-1 finalisation, as workgroup roles -- [0, r[0].nblk) reduce conv3's partials, the next r[1].nblk conv2's, the rest
// finalise layer 1.  The reduces used to sit between the GEMMs of the chain (wgrad<3> -> reduce -> wgrad<2> -> reduce ->
// finalise): three launches on the critical chain of every scale that nothing else waited for.
struct PnTailArgs {
    ReduceJob r[2];
    L1Args l1;
    int has_l1;
};
__global__ __launch_bounds__(WR_T) void pn_tail_kernel(PnTailArgs t)
{
    __shared__ __attribute__((aligned(16))) float sh[WR_T * 4];
    const int bid = (int)blockIdx.x;
    if (bid < t.r[0].nblk) wgrad_reduce_body(t.r[0], bid, sh);
    else if (bid < t.r[0].nblk + t.r[1].nblk) wgrad_reduce_body(t.r[1], bid - t.r[0].nblk, sh);
    else if (t.has_l1) l1_finalize_body(t.l1, (bid - t.r[0].nblk - t.r[1].nblk) * WR_T + (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
template <int LAYER, int DZ3 = 0>
static int launch_dgrad(const DgradArgs &a, int B, int precision, hipStream_t st)
{
    if (a.CRED % 64 || a.CPREV % 64 || a.CRED > MAXC) return FCN_E_BADARG;
    if ((int64_t)B * a.cap * (a.CRED > a.CPREV ? a.CRED : a.CPREV) >= (int64_t)1 << 31) return FCN_E_LIMIT;   // 32-bit offsets
    const unsigned nt = (unsigned)(B * a.tps);
#if FCN_DG_WIDE     // 64 x 256 tiles (8 waves): the dy rows of a tile are built ONCE for all 256 columns.  Rounds 3 / 4 measured it equal
                    // (231.4 vs 230.2 us for the widest scale's backward) and left it out; behind round 5's staging diet -- the
                    // operand transform is what is left of the loop's vector work -- it is 1.0-1.2 % of the step (EXPERIMENTS 5.6)
    if (a.CPREV % 256 == 0) {
        FCN_MM_SWITCH(FCN_MM_OF(precision, false),
                      hipLaunchKernelGGL((dgrad_kernel<MM, LAYER, 1, 2, 4, DZ3>), dim3((2 * nt * (a.CPREV / 256) + 7) / 8 * 8), dim3(512), 0, st, a));
        FCN_CHECK_LAUNCH();
        return 0;
    }
#endif
    if (a.CPREV % 128 == 0) {          // 64 x 128 tiles, two workgroups per listed 128-row tile
        FCN_MM_SWITCH(FCN_MM_OF(precision, false),
                      hipLaunchKernelGGL((dgrad_kernel<MM, LAYER, 1, 2, 2, DZ3>), dim3((2 * nt * (a.CPREV / 128) + 7) / 8 * 8), dim3(256), 0, st, a));
    } else {                            // 64 x 64 tiles (the 64-channel layers of scales 1 and 2: few column tiles)
        FCN_MM_SWITCH(FCN_MM_OF(precision, false),
                      hipLaunchKernelGGL((dgrad_kernel<MM, LAYER, 1, 1, 2, DZ3>), dim3((2 * nt * (a.CPREV / 64) + 7) / 8 * 8), dim3(256), 0, st, a));
    }
    FCN_CHECK_LAUNCH();
    return 0;
}

template <int MM, int LAYER, int RC = 0>
static void launch_wgrad_mm(const WgradArgs &a, dim3 grid, bool m2, bool n2, hipStream_t st)
{
    if (m2 && n2) hipLaunchKernelGGL((wgrad_kernel<MM, LAYER, 2, 2, RC>), grid, dim3(GT), 0, st, a);
    else if (m2) hipLaunchKernelGGL((wgrad_kernel<MM, LAYER, 2, 1, RC>), grid, dim3(GT), 0, st, a);
    else if (n2) hipLaunchKernelGGL((wgrad_kernel<MM, LAYER, 1, 2, RC>), grid, dim3(GT), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<MM, LAYER, 1, 1, RC>), grid, dim3(GT), 0, st, a);
}

struct WgradPlan {
    int nsplit, oy, oz;
    bool m2, n2;
};

// split count and tile grid of a weight-gradient GEMM
template <int LAYER>
static int plan_wgrad(const WgradArgs &a, int B, int nsplit_cap, WgradPlan &P)
{
    P.m2 = (a.COUT % 128 == 0); P.n2 = (a.CIN % 128 == 0);
    P.oy = a.COUT / (P.m2 ? 128 : 64); P.oz = a.CIN / (P.n2 ? 128 : 64);
    // FCN_WG_SLOTS workgroup slots per launch (two thirds for the 128 x 128 tile of layer 2), swept on MI355X over 256 / 384 / 768 / 1536:
    // every split writes a full (COUT, CIN) partial that the reduce reads back (at 768 slots 150 MB written + read per
    // step), and fewer, longer splits amortise the per-workgroup prologue -- 768 -> 384 took 14 us off the PointNet backward.
    // Each split takes ceil(live_tiles / nsplit) row tiles (computed on the device, where the live count is known), so
    // splits stay balanced whatever the occupancy of the frustums.
#ifndef FCN_WG_SLOTS
#define FCN_WG_SLOTS 256      // (re-swept in round 3 with the replicated sum slots: 384 -> 1.3587, 256 -> 1.3548, 512 -> 1.3630 ms per step)
#endif
    const int slots = (LAYER == 2 && P.m2 && P.n2) ? (FCN_WG_SLOTS * 2) / 3 : FCN_WG_SLOTS;
    if ((int64_t)B * a.cap * (a.COUT > a.CIN ? a.COUT : a.CIN) >= (int64_t)1 << 31) return FCN_E_LIMIT;   // 32-bit offsets
    if ((int64_t)B * a.cap >= (int64_t)1 << 24) return FCN_E_LIMIT;                                        // 24-bit row multiplies
    int nsplit = slots / (P.oy * P.oz);
    if (nsplit < (B * a.tps + WG_TMAX - 1) / WG_TMAX) nsplit = (B * a.tps + WG_TMAX - 1) / WG_TMAX;      // tiles per split <= WG_TMAX
    if (nsplit < 1) nsplit = 1;
    if (nsplit > B * a.tps) nsplit = B * a.tps;
    if (nsplit > nsplit_cap) nsplit = nsplit_cap;
    P.nsplit = nsplit;
    return 0;
}

// the fixed-order sum of the split partials into the torch weight layout
static int make_reduce_job(const WgradArgs &a, const WgradPlan &P, float *out, ReduceJob &j)
{
    const int64_t ne = (int64_t)a.COUT * a.CIN;
    if (((uintptr_t)out & 15) != 0) return FCN_E_BADARG;        // the reduce writes 16-byte vectors (include/fcn_hip.h)
    const int nsplit = P.nsplit;
    const int gr = nsplit >= 128 ? 16 : (nsplit >= 64 ? 8 : (nsplit >= 32 ? 4 : (nsplit >= 16 ? 2 : 1)));
    j.partial = a.partial; j.tiles = a.tiles; j.out = out; j.nelem = ne; j.nsplit = nsplit; j.gr = gr;
    j.nblk = (int)(ne / (4 * (WR_T / gr)));
    return 0;
}

static int launch_wgrad_reduce(const WgradArgs &a, const WgradPlan &P, hipStream_t st, float *out)
{
    ReduceJob j;
    FCN_TRY(make_reduce_job(a, P, out, j));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)j.nblk), dim3(WR_T), 0, st, j);
    FCN_CHECK_LAUNCH();
    return 0;
}

static L1Args make_l1(const fcn_pn_desc *d, const fcn_pn_params *p, const fcn_pn_ws *ws, double *bsQ, int brs, const float *bn1, double M,
                      float *dW0, float *dgamma0, float *dbeta0)
{
    L1Args a;
    a.Qr = bsQ; a.rep_stride = brs; a.mom = ws->stat + FCN_STAT_MOM; a.W1 = p->W[0]; a.gamma = p->gamma[0]; a.bn1 = bn1; a.C = d->C1; a.M = M;
    a.dW1 = dW0; a.dgamma = dgamma0; a.dbeta = dbeta0;
    return a;
}

static int launch_l1(const L1Args &a, hipStream_t st)
{
    hipLaunchKernelGGL(l1_finalize_kernel, dim3((a.C + 63) / 64), dim3(64), 0, st, a);
    FCN_CHECK_LAUNCH();
    return 0;
}

// both reduces (+ the layer-1 finalisation) of a scale in one launch (pn_tail_kernel)
static int launch_tail(const ReduceJob &r3, const ReduceJob &r2, const L1Args *l1, hipStream_t st)
{
    PnTailArgs t;
    t.r[0] = r3; t.r[1] = r2;
    t.has_l1 = l1 ? 1 : 0;
    if (l1) t.l1 = *l1; else t.l1 = L1Args();
    const unsigned nb = (unsigned)(r3.nblk + r2.nblk + (l1 ? (l1->C + WR_T - 1) / WR_T : 0));
    hipLaunchKernelGGL(pn_tail_kernel, dim3(nb), dim3(WR_T), 0, st, t);
    FCN_CHECK_LAUNCH();
    return 0;
}

// rewait: an event the stream waits for AGAIN between the GEMM and its reduce (fcn_pn_backward3: a redundant edge that steers ROCm's
// graph executor -- see pn_backward_impl)
// defer: the reduce is not launched -- *defer receives its job (the caller runs it in a tail launch)
template <int LAYER>
static int launch_wgrad(WgradArgs &a, int B, int nsplit_cap, int precision, hipStream_t st, float *out, hipEvent_t rewait = nullptr,
                        ReduceJob *defer = nullptr)
{
    WgradPlan P;
    FCN_TRY(plan_wgrad<LAYER>(a, B, nsplit_cap, P));
    dim3 grid(P.nsplit, P.oy, P.oz);
    if (LAYER == 3 && !a.dy) {             // dy3 rebuilt by the kernel (RC)
        if (!a.ycur || !a.ewin || !a.amax || !a.gmax || !a.cb.bstat) return FCN_E_BADARG;
        FCN_MM_SWITCH(FCN_MM_OF(precision, false), (launch_wgrad_mm<MM, LAYER, LAYER == 3 ? 1 : 0>(a, grid, P.m2, P.n2, st)));
    } else {
        FCN_MM_SWITCH(FCN_MM_OF(precision, false), (launch_wgrad_mm<MM, LAYER>(a, grid, P.m2, P.n2, st)));
    }
    FCN_CHECK_LAUNCH();
    if (defer) return make_reduce_job(a, P, out, *defer);
    if (rewait) {
        const hipError_t e = hipStreamWaitEvent(st, rewait, 0);
        if (e != hipSuccess) return (int)e;
    }
    return launch_wgrad_reduce(a, P, st, out);
}

// conv2's data gradient + both weight gradients of a scale in ONE launch (pn_mid_kernel).  Supported tile plans: the all-64 one
// (C1 = C2 = 64, C3 = 128: the narrow scales) and the all-128 one (every width a multiple of 128); -1: not supported, run them apart.
static int launch_mid(const DgradArgs &g2, const WgradArgs &w3, const WgradArgs &w2, const WgradPlan &P3, const WgradPlan &P2,
                      int B, int precision, hipStream_t st)
{
    if (g2.CRED % 64 || g2.CPREV % 64 || g2.CRED > MAXC) return FCN_E_BADARG;
    if ((int64_t)B * g2.cap * (g2.CRED > g2.CPREV ? g2.CRED : g2.CPREV) >= (int64_t)1 << 31) return FCN_E_LIMIT;
    const bool d2 = g2.CPREV % 128 == 0;
    PnMidArgs m;
    m.g2 = g2; m.w3 = w3; m.w2 = w2;
    const unsigned nt = (unsigned)(B * g2.tps);
    m.n_g2 = (int)((2 * nt * (g2.CPREV / (d2 ? 128 : 64)) + 7) / 8 * 8);
    m.ns3 = P3.nsplit; m.oy3 = P3.oy; m.n_w3 = P3.nsplit * P3.oy * P3.oz;
    m.ns2 = P2.nsplit; m.oy2 = P2.oy;
    const unsigned grid = (unsigned)(m.n_g2 + m.n_w3 + P2.nsplit * P2.oy * P2.oz);
    const bool all128 = d2 && P3.m2 && P3.n2 && P2.m2 && P2.n2;
    const bool all64 = !d2 && P3.m2 && !P3.n2 && !P2.m2 && !P2.n2;
    if (all128) {
        FCN_MM_SWITCH(FCN_MM_OF(precision, false), hipLaunchKernelGGL((pn_mid_kernel<MM, 2, 2, 2, 2, 2>), dim3(grid), dim3(GT), 0, st, m));
    } else if (all64) {
        FCN_MM_SWITCH(FCN_MM_OF(precision, false), hipLaunchKernelGGL((pn_mid_kernel<MM, 1, 2, 1, 1, 1>), dim3(grid), dim3(GT), 0, st, m));
    } else {
        return -1;
    }
    FCN_CHECK_LAUNCH();
    return 0;
}

extern "C" int fcn_pn_backward2(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                                const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                                void *stream, void *stream2, void *const *events);

extern "C" int fcn_pn_backward(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                               const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                               void *stream)
{
    return fcn_pn_backward2(d, p, dfeat, ws, dW, dgamma, dbeta, stream, nullptr, nullptr);
}

// Same with the weight gradients (wgrad + reduce of conv3 and conv2) on a second stream beside the data-gradient
// chain poolbwd -> dgrad3 -> dgrad2 -> l1_finalize.  events: 3 caller-owned hipEvent_t (fork, fork, join).
static int pn_backward_impl(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat, const fcn_pn_ws *ws,
                            float *dW[3], float *dgamma[3], float *dbeta[3], void *stream, void *stream2, void *stream3,
                            void *const *events, const float *dz3_dense = nullptr);

// Backward of the UN-POOLED module output (PointNetModule.forward's (B, C3, L, K) return, models/det_base.py:62-103): dz3 (B,cap,C3)
// is the gradient w.r.t. relu(bn3(y3)) of every ENTRY row -- the K slots of a window summed back onto its rows (the first hit
// collects its K - ne + 1 duplicates), times the ReLU mask -- and ws->bstat replica 0 holds sum dz3 [C3], sum dz3 * xhat3 [C3]
// (the other replicas zero), both prepared by the caller.  Everything behind that is the pooled backward's chain.
extern "C" int fcn_pn_backward_dense(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dz3, const fcn_pn_ws *ws,
                                     float *dW[3], float *dgamma[3], float *dbeta[3], void *stream)
{
    if (!dz3) return FCN_E_BADARG;
    return pn_backward_impl(d, p, dz3, ws, dW, dgamma, dbeta, stream, nullptr, nullptr, nullptr, dz3);
}

extern "C" int fcn_pn_backward2(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                                const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                                void *stream, void *stream2, void *const *events)
{
    return pn_backward_impl(d, p, dfeat, ws, dW, dgamma, dbeta, stream, stream2, nullptr, events);
}

// Three streams: after dgrad3 (which leaves dy3, dz2 and the BN2-backward sums final) the chain splits three ways -- dgrad2 +
// l1_finalize on `stream`, conv3's weight gradient (+ reduce) on `stream2`, conv2's on `stream3` -- instead of two: on the
// widest scale the second stream's wgrad3 -> reduce -> wgrad2 -> reduce (121 us) was longer than the main chain's rest (67 us)
// and set the scale's backward (232 us isolated).  Needs ws->partial sized for BOTH weight gradients at once
// (nsplit * (C3*C2 + C2*C1) floats: conv2's partials follow conv3's) and 4 events (fork, unused, join 2, join 3).
extern "C" int fcn_pn_backward3(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                                const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                                void *stream, void *stream2, void *stream3, void *const *events)
{
    if (!stream2 || !stream3 || !events) return FCN_E_BADARG;
    return pn_backward_impl(d, p, dfeat, ws, dW, dgamma, dbeta, stream, stream2, stream3, events);
}

static int pn_backward_impl(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat, const fcn_pn_ws *ws,
                            float *dW[3], float *dgamma[3], float *dbeta[3], void *stream, void *stream2, void *stream3,
                            void *const *events, const float *dz3_dense)
{
    if (!d || !p || !ws || !dfeat || !dW || !dgamma || !dbeta) return FCN_E_BADARG;
    if (!d->training || !ws->wenc) return FCN_E_BADARG;
    if (d->precision < 0 || d->precision > FCN_PREC_BF16_OPS) return FCN_E_BADARG;
    if (ws->partial_both < 0 || ws->partial_both > 2) return FCN_E_BADARG;      // (an uninitialised trailing field must not enable the merged launches)
    hipStream_t st = (hipStream_t)stream;
    const bool two = stream2 != nullptr && events != nullptr;
    const bool three = two && stream3 != nullptr;
    hipStream_t sw = two ? (hipStream_t)stream2 : st;
    hipStream_t sw2 = three ? (hipStream_t)stream3 : sw;        // stream of conv2's weight gradient
    const int B = d->B, L = d->L, K = d->K, C1 = d->C1, C2 = d->C2, C3 = d->C3;
    if (C1 % 64 || C2 % 64 || C3 % 64 || C1 > MAXC || C2 > MAXC || C3 > MAXC) return FCN_E_BADARG;
    const int cap = L * K;
    const double M = (double)B * (double)L * (double)K;
    const int tps = (cap + 127) / 128;
    if (ws->nsplit < B * tps) return FCN_E_BADARG;
    const float *bn1 = ws->bn + fcn_bn_off(0, C1, C2);
    const float *bn2 = ws->bn + fcn_bn_off(1, C1, C2);
    const float *bn3 = ws->bn + fcn_bn_off(2, C1, C2);
    const int brs = 2 * C3 + 2 * C2 + 4 * C1;        // doubles per replica block of ws->bstat
    double *bs3 = ws->bstat, *bs2 = bs3 + 2 * C3, *bsQ = bs2 + 2 * C2;

    hipError_t e = hipSuccess;          // ws->bstat was zeroed by the pool kernel of this scale's forward

    if (dz3_dense) {
        if (!ws->dy3 || d->precision == FCN_PREC_BF16) return FCN_E_BADARG;      // (dz3 is fp32 rows; the dense form keeps dy3 materialised)
    } else if (d->precision == FCN_PREC_BF16)       // y3 stored as bf16 (gemm_tile.h: St)
        hipLaunchKernelGGL(poolbwd_kernel<1>, dim3((L + PWB - 1) / PWB, C3 / 64, B), dim3(GT), 0, st, dfeat, ws->amax,
                           ws->y3, bn3, ws->gmax, bs3, brs, L, cap, C3, C3 + d->nvec, d->nlc, (int)fcn_pn_key_pool(d, ws, C3));
    else
        hipLaunchKernelGGL(poolbwd_kernel<0>, dim3((L + PWB - 1) / PWB, C3 / 64, B), dim3(GT), 0, st, dfeat, ws->amax,
                           ws->y3, bn3, ws->gmax, bs3, brs, L, cap, C3, C3 + d->nvec, d->nlc, (int)fcn_pn_key_pool(d, ws, C3));
    FCN_CHECK_LAUNCH();
    DgradArgs g;
    g.ent = (const float4 *)ws->ent; g.woff = ws->woff; g.tiles = ws->tiles; g.ewin = ws->ewin; g.L = L; g.cap = cap; g.tps = tps;
    g.ycur = ws->y3; g.amax = ws->amax; g.gmax = ws->gmax; g.dzcur = nullptr;
    g.Wenc = (const u32x4 *)(ws->wenc + 2 * (int64_t)C2 * C1 + (int64_t)C3 * C2);            // G3 (pn_wenc_off(3))
    g.cb.bstat = bs3; g.cb.rep_stride = brs; g.cb.gamma = p->gamma[2]; g.cb.bn = bn3; g.cb.invM = 1.0 / M; g.cb.dgamma = dgamma[2]; g.cb.dbeta = dbeta[2];
    g.dybuf = ws->dy3; g.yprev = ws->y2; g.bn_prev = bn2; g.W1 = nullptr; g.dzprev = ws->dz2; g.bstat_prev = bs2;
    g.CRED = C3; g.CPREV = C2;
#if FCN_XB & 512       // (timing build: the cost of a2 . G instead of dy3 . W3 -- reduction over C2, the A operand read from y2)
    g.ycur = ws->y2; g.CRED = C2;
#endif
    if (dz3_dense) {
        g.dzcur = dz3_dense; g.amax = nullptr; g.gmax = nullptr;
        FCN_TRY((launch_dgrad<3, 1>(g, B, d->precision, st)));
    } else {
        FCN_TRY(launch_dgrad<3>(g, B, d->precision, st));
    }

    WgradArgs w;
    w.ent = (const float4 *)ws->ent; w.woff = ws->woff; w.tiles = ws->tiles; w.L = L; w.cap = cap; w.tps = tps;
    w.partial = ws->partial;
    w.dy = ws->dy3; w.dz = nullptr; w.ycur = nullptr; w.yprev = ws->y2; w.bn_prev = bn2;
    w.cb.bstat = nullptr; w.cb.rep_stride = brs; w.cb.gamma = nullptr; w.cb.bn = nullptr; w.cb.invM = 1.0 / M; w.cb.dgamma = nullptr; w.cb.dbeta = nullptr;
    w.ewin = nullptr; w.amax = nullptr; w.gmax = nullptr;
    if (!ws->dy3) {        // no dy3 buffer: conv3's weight-gradient GEMM rebuilds dy3 from what the data-gradient GEMM reads
        w.ycur = ws->y3; w.cb.bstat = bs3; w.cb.gamma = p->gamma[2]; w.cb.bn = bn3;
        w.ewin = ws->ewin; w.amax = ws->amax; w.gmax = ws->gmax;
    }
    w.W1 = nullptr; w.COUT = C3; w.CIN = C2;
#if FCN_XB & 256       // (timing build: the cost of the Gram matrix a2^T a2 instead of dy3^T a2)
    w.dy = ws->y2; w.COUT = C2;
#endif
    // partial_both: `partial` holds both weight gradients' partials at once -- 1: the merged middle launch (one stream), 2: separate
    // GEMMs; either way the two reduces (+ the layer-1 finalisation where it is on the same stream) run as ONE tail launch
    const bool both = ws->partial_both != 0;
    const L1Args l1a = make_l1(d, p, ws, bsQ, brs, bn1, M, dW[0], dgamma[0], dbeta[0]);
    if (!two && ws->partial_both == 1 && ws->dy3) {
        // ONE stream: conv2's data gradient and both weight gradients ride in one launch (they depend on dgrad<3> only), then the
        // two reduces and the layer-1 finalisation -- 6 launches per scale instead of 8, none of them waiting for a sibling
        DgradArgs g2 = g;
        g2.ycur = ws->y2; g2.amax = nullptr; g2.gmax = nullptr; g2.dzcur = ws->dz2;
        g2.Wenc = (const u32x4 *)(ws->wenc + (int64_t)C2 * C1 + (int64_t)C3 * C2);               // G2 (pn_wenc_off(2))
        g2.cb.bstat = bs2; g2.cb.gamma = p->gamma[1]; g2.cb.bn = bn2; g2.cb.dgamma = dgamma[1]; g2.cb.dbeta = dbeta[1];
        g2.dybuf = nullptr; g2.yprev = nullptr; g2.bn_prev = bn1; g2.W1 = p->W[0]; g2.dzprev = nullptr; g2.bstat_prev = bsQ;
        g2.CRED = C2; g2.CPREV = C1;
        WgradArgs w2 = w;
        w2.dy = nullptr; w2.dz = ws->dz2; w2.ycur = ws->y2; w2.yprev = nullptr; w2.bn_prev = bn1;
        w2.cb.bstat = bs2; w2.cb.gamma = p->gamma[1]; w2.cb.bn = bn2;
        w2.W1 = p->W[0]; w2.COUT = C2; w2.CIN = C1;
        w2.partial = ws->partial + (int64_t)ws->nsplit * C3 * C2;      // its own partials: the two weight gradients run at once
        WgradPlan P3, P2;
        FCN_TRY(plan_wgrad<3>(w, B, ws->nsplit, P3));
        FCN_TRY(plan_wgrad<2>(w2, B, ws->nsplit, P2));
        const int rc = launch_mid(g2, w, w2, P3, P2, B, d->precision, st);
        if (rc > 0) return rc;
        if (rc == 0) {
            ReduceJob r3, r2;
            FCN_TRY(make_reduce_job(w, P3, dW[2], r3));
            FCN_TRY(make_reduce_job(w2, P2, dW[1], r2));
            return launch_tail(r3, r2, &l1a, st);
        }                               // (rc < 0: a tile plan the merged launch has no instance for -- the launches below)
    }
    if (two) {     // dy3 is final: conv3's weight gradient can run beside the rest of the chain
        e = hipEventRecord((hipEvent_t)events[0], st);
        if (e != hipSuccess) return (int)e;
        e = hipStreamWaitEvent(sw, (hipEvent_t)events[0], 0);
        if (e != hipSuccess) return (int)e;
        if (three) {
            e = hipStreamWaitEvent(sw2, (hipEvent_t)events[0], 0);
            if (e != hipSuccess) return (int)e;
        }
    }
    // three streams: ROCm 7.2's graph executor gives child i of a node on its internal stream p the stream (p + i) % 4, counting every
    // edge out of the node.  dgrad<3>'s edges in capture order: conv3's weight-gradient GEMM (0: stays on p), its reduce through a SECOND
    // wait for events[0] (1: redundant, the node is already placed), conv2's data gradient (2) and conv2's weight gradient (3) -- so the
    // two branches land on the streams of the third and fourth captured scale (the narrow ones), not on the second's (tools/graph_dot.py)
    ReduceJob r3, r2;
    const bool tail = both && !three;         // the reduces deferred to one launch behind conv2's weight-gradient GEMM
    FCN_TRY(launch_wgrad<3>(w, B, ws->nsplit, d->precision, sw, dW[2], three ? (hipEvent_t)events[0] : nullptr, tail ? &r3 : nullptr));
    if (three) {
        e = hipEventRecord((hipEvent_t)events[3], sw);
        if (e != hipSuccess) return (int)e;
        // conv2's data gradient + the layer-1 finalisation FIRST (edge 2), conv2's weight gradient after them (edge 3)
        DgradArgs g2 = g;
        g2.ycur = ws->y2; g2.amax = nullptr; g2.gmax = nullptr; g2.dzcur = ws->dz2;
        g2.Wenc = (const u32x4 *)(ws->wenc + (int64_t)C2 * C1 + (int64_t)C3 * C2);
        g2.cb.bstat = bs2; g2.cb.gamma = p->gamma[1]; g2.cb.bn = bn2; g2.cb.dgamma = dgamma[1]; g2.cb.dbeta = dbeta[1];
        g2.dybuf = nullptr; g2.yprev = nullptr; g2.bn_prev = bn1; g2.W1 = p->W[0]; g2.dzprev = nullptr; g2.bstat_prev = bsQ;
        g2.CRED = C2; g2.CPREV = C1;
        FCN_TRY(launch_dgrad<2>(g2, B, d->precision, st));
        FCN_TRY(launch_l1(l1a, st));
        w.dy = nullptr; w.dz = ws->dz2; w.ycur = ws->y2; w.yprev = nullptr; w.bn_prev = bn1;
        w.cb.bstat = bs2; w.cb.gamma = p->gamma[1]; w.cb.bn = bn2;
        w.W1 = p->W[0]; w.COUT = C2; w.CIN = C1;
        w.partial = ws->partial + (int64_t)ws->nsplit * C3 * C2;      // its own partials: the two weight gradients run at once
        FCN_TRY(launch_wgrad<2>(w, B, ws->nsplit, d->precision, sw2, dW[1]));
        e = hipEventRecord((hipEvent_t)events[2], sw2);
        if (e != hipSuccess) return (int)e;
        e = hipStreamWaitEvent(st, (hipEvent_t)events[2], 0);
        if (e != hipSuccess) return (int)e;
        e = hipStreamWaitEvent(st, (hipEvent_t)events[3], 0);
        if (e != hipSuccess) return (int)e;
        return 0;
    }

    if (two) {     // dz2 and its BN-backward sums were final at events[0]: conv2's weight gradient follows conv3's on the side
                   // stream (two streams) or runs beside it on the third
        if (!three) {
            e = hipEventRecord((hipEvent_t)events[1], st);
            if (e != hipSuccess) return (int)e;
            e = hipStreamWaitEvent(sw, (hipEvent_t)events[1], 0);
            if (e != hipSuccess) return (int)e;
        }
        w.dy = nullptr; w.dz = ws->dz2; w.ycur = ws->y2; w.yprev = nullptr; w.bn_prev = bn1;
        w.cb.bstat = bs2; w.cb.gamma = p->gamma[1]; w.cb.bn = bn2;
        w.W1 = p->W[0]; w.COUT = C2; w.CIN = C1;
        if (three || tail) w.partial = ws->partial + (int64_t)ws->nsplit * C3 * C2;      // its own partials
        FCN_TRY(launch_wgrad<2>(w, B, ws->nsplit, d->precision, sw2, dW[1], nullptr, tail ? &r2 : nullptr));
        if (tail) FCN_TRY(launch_tail(r3, r2, nullptr, sw2));       // (the layer-1 finalisation ends the OTHER stream's chain)
        e = hipEventRecord((hipEvent_t)events[2], sw2);
        if (e != hipSuccess) return (int)e;
    }
    g.ycur = ws->y2; g.amax = nullptr; g.gmax = nullptr; g.dzcur = ws->dz2;
    g.Wenc = (const u32x4 *)(ws->wenc + (int64_t)C2 * C1 + (int64_t)C3 * C2);                // G2 (pn_wenc_off(2))
    g.cb.bstat = bs2; g.cb.gamma = p->gamma[1]; g.cb.bn = bn2; g.cb.dgamma = dgamma[1]; g.cb.dbeta = dbeta[1];
    g.dybuf = nullptr; g.yprev = nullptr; g.bn_prev = bn1; g.W1 = p->W[0]; g.dzprev = nullptr; g.bstat_prev = bsQ;
    g.CRED = C2; g.CPREV = C1;
    FCN_TRY(launch_dgrad<2>(g, B, d->precision, st));

    if (!two) {
        w.dy = nullptr; w.dz = ws->dz2; w.ycur = ws->y2; w.yprev = nullptr; w.bn_prev = bn1;
        w.cb.bstat = bs2; w.cb.gamma = p->gamma[1]; w.cb.bn = bn2;
        w.W1 = p->W[0]; w.COUT = C2; w.CIN = C1;
        if (tail) w.partial = ws->partial + (int64_t)ws->nsplit * C3 * C2;
        FCN_TRY(launch_wgrad<2>(w, B, ws->nsplit, d->precision, st, dW[1], nullptr, tail ? &r2 : nullptr));
        if (tail) return launch_tail(r3, r2, &l1a, st);
    }

    FCN_TRY(launch_l1(l1a, st));
    if (two) {
        e = hipStreamWaitEvent(st, (hipEvent_t)events[2], 0);
        if (e != hipSuccess) return (int)e;
        if (three) {
            e = hipStreamWaitEvent(st, (hipEvent_t)events[3], 0);
            if (e != hipSuccess) return (int)e;
        }
    }
    return 0;
}
