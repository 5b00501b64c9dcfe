// TUNING-BUILD HOOKS of libfcn_hip.so, all in one place.  Nothing in this file is active in the product: every mask below is 0 and
// every probe macro expands to nothing unless a tuning build (tools/build_variant.py <name> -D...) defines it -- the build hash
// (fcn_build_hash) covers the flags, so a tuning build can never pass for the product.  The kernel sources only hold the call
// sites: `if (FCN_X & bit)` around a phase (an ablation: results are WRONG by construction when a bit is set, the number of
// interest is the kernel's duration without that phase) and `PNF_ADD / PNP_ADD / PROBE_STAMP` between phases (cycle or clock
// accounting by wave 0 of every workgroup, read back by tools/pn_probe.py / fcn_probe.py / fcn_probe_bwd.py).
#pragma once

// ---- ablation masks (timing experiments; EXPERIMENTS.md rounds 2-4 hold what they measured) ------------------------------------
// FCN_EXP (gemm_tile.h operand encoders): bit 0 stores the weight
// operand without encoding, bit 1 the activation / gradient operand -- upper bounds for what pre-encoded operands could buy.
#ifndef FCN_EXP
#define FCN_EXP 0
#endif
#define MM_ENC_W ((FCN_EXP & 1) ? MM_F32 : MM)
#define MM_ENC_A ((FCN_EXP & 2) ? MM_F32 : MM)

// FCN_X (pointnet_fwd.hip, fwd_gemm_kernel): bits -- 1: A loaded for the first chunk only,
// 2: W loaded for the first chunk only, 4: LDS staging for the first chunk only, 8: no MFMAs, 16: no output stores / statistics,
// 32: no output stores (statistics kept), 64: no statistics atomics
#ifndef FCN_X
#define FCN_X 0
#endif

// FCN_XB (pointnet_bwd.hip, dgrad_kernel): bits -- 1: A-side loads for the first
// chunk only, 2: W loads first chunk only, 4: staging first chunk only, 8: no MFMAs, 16: no epilogue, 32: no dz stores, 64: no
// statistics atomics, 128: no dy3 store
#ifndef FCN_XB
#define FCN_XB 0
#endif

// FCN_XF (fcn_net.hip, forward K-group kernel): bits -- 1: activation
// loads for a group's first chunk only, 2: weight loads first chunk only, 4: LDS staging first chunk only, 8: no MFMAs,
// 16: no epilogue (cross-group sum, stores, statistics), 32: no BN prologue (scale 1 / shift 0), 64: no statistics atomics,
// 128: return at entry (bare launches), 256: return behind the prologue
#ifndef FCN_XF
#define FCN_XF 0
#endif
// FCN_XG: the same for the backward roles -- 1: dz / y loads of a data-gradient tile for a group's first chunk only, 2: weight loads
// first chunk only, 4: staging first chunk only, 8: no MFMAs, 16: no data-gradient epilogue; 256 / 512 / 1024: the same as 1+2 / 4 /
// 8 in the weight-gradient role (CGB_NO_WGRAD / CGB_NO_REDUCE / CGB_NO_DGRAD drop whole roles)
#ifndef FCN_XG
#define FCN_XG 0
#endif

// ---- probes: device tables + readers live in the ONE source that defines the matching FCN_TUNING_* switch before including this ----
#ifdef FCN_TUNING_PNF
// Intra-kernel cycle accounting of the forward GEMM for TUNING BUILDS ONLY (-DFCN_PROBE, tools/pn_probe.py fwd; never compiled
// into the product): wave 0 of every workgroup sums the shader-clock cycles it spends in each phase of the K loop.
#ifdef FCN_PROBE
#define PNF_MAX 32768
__device__ unsigned long long g_pnf_probe[PNF_MAX * 8];
__device__ unsigned int g_pnf_probe_n;
#define PNF_DECL unsigned long long pa_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long pt_ = clock64(), pt0_ = pt_
#define PNF_ADD(i) do { const unsigned long long n_ = clock64(); pa_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PNF_FLUSH(tag)                                                                                   \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            const unsigned int s_ = atomicAdd(&g_pnf_probe_n, 1u);                                       \
            if (s_ < PNF_MAX) {                                                                          \
                g_pnf_probe[s_ * 8] = (unsigned long long)(tag);                                         \
                g_pnf_probe[s_ * 8 + 1] = clock64() - pt0_;                                              \
                for (int q_ = 0; q_ < 6; ++q_) g_pnf_probe[s_ * 8 + 2 + q_] = pa_[q_];                   \
            }                                                                                            \
        }                                                                                                \
    } while (0)
extern "C" int fcn_pn_probe_read_fwd(unsigned long long *host_out, int max_records, int reset)
{
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_pnf_probe_n), sizeof(n)) != hipSuccess) return -1;
    if ((int)n > max_records) n = max_records;
    if (n > PNF_MAX) n = PNF_MAX;
    if (n && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pnf_probe), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_pnf_probe_n), &z, sizeof(z)); }
    return (int)n;
}
#else
#define PNF_DECL
#define PNF_ADD(i)
#define PNF_FLUSH(tag)
#endif
#endif

#ifdef FCN_TUNING_PNP
// Intra-kernel cycle accounting of the data-gradient GEMM for TUNING BUILDS ONLY (-DFCN_PROBE, tools/pn_probe.py; never
// compiled into the product): wave 0 of every workgroup sums the shader-clock cycles it spends in each phase of the K loop.
#ifdef FCN_PROBE
#define PNP_MAX 32768
__device__ unsigned long long g_pn_probe[PNP_MAX * 8];
__device__ unsigned int g_pn_probe_n;
#define PNP_DECL unsigned long long pa_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long pt_ = clock64(), pt0_ = pt_
#define PNP_ADD(i) do { const unsigned long long n_ = clock64(); pa_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PNP_FLUSH(tag)                                                                                   \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            const unsigned int s_ = atomicAdd(&g_pn_probe_n, 1u);                                        \
            if (s_ < PNP_MAX) {                                                                          \
                g_pn_probe[s_ * 8] = (unsigned long long)(tag);                                          \
                g_pn_probe[s_ * 8 + 1] = clock64() - pt0_;                                               \
                for (int q_ = 0; q_ < 6; ++q_) g_pn_probe[s_ * 8 + 2 + q_] = pa_[q_];                    \
            }                                                                                            \
        }                                                                                                \
    } while (0)
extern "C" int fcn_pn_probe_read(unsigned long long *host_out, int max_records, int reset)
{
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_pn_probe_n), sizeof(n)) != hipSuccess) return -1;
    if ((int)n > max_records) n = max_records;
    if (n > PNP_MAX) n = PNP_MAX;
    if (n && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pn_probe), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_pn_probe_n), &z, sizeof(z)); }
    return (int)n;
}
#else
#define PNP_DECL
#define PNP_ADD(i)
#define PNP_FLUSH(tag)
#endif
#endif

#ifdef FCN_TUNING_FCN
// Intra-kernel phase stamps for TUNING BUILDS ONLY (-DFCN_PROBE, tools/fcn_probe.py; never compiled into the product):
// wave 0 of every workgroup records the 100 MHz device clock at phase boundaries into a global table.
#ifdef FCN_PROBE
#define FCN_PROBE_MAX 65536
__device__ unsigned long long g_fcn_probe[FCN_PROBE_MAX * 8];
__device__ unsigned int g_fcn_probe_n;
#define PROBE_DECL unsigned long long pb_[8]; int pbn_ = 0
#define PROBE_STAMP() do { if (pbn_ < 7) pb_[pbn_++] = wall_clock64(); } while (0)
#define PROBE_FLUSH(tag)                                                                                 \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            const unsigned int s_ = atomicAdd(&g_fcn_probe_n, 1u);                                       \
            if (s_ < FCN_PROBE_MAX) {                                                                    \
                g_fcn_probe[s_ * 8] = (unsigned long long)(tag);                                         \
                for (int q_ = 0; q_ < 7; ++q_) g_fcn_probe[s_ * 8 + 1 + q_] = q_ < pbn_ ? pb_[q_] : 0ull; \
            }                                                                                            \
        }                                                                                                \
    } while (0)
extern "C" int fcn_probe_read(unsigned long long *host_out, int max_records, int reset)
{
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fcn_probe_n), sizeof(n)) != hipSuccess) return -1;
    if ((int)n > max_records) n = max_records;
    if (n > FCN_PROBE_MAX) n = FCN_PROBE_MAX;
    if (n && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fcn_probe), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_fcn_probe_n), &z, sizeof(z)); }
    return (int)n;
}
#else
#define PROBE_DECL
#define PROBE_STAMP()
#define PROBE_FLUSH(tag)
#endif
// (-DFCN_PROBE=3: the stamps of the BACKWARD roles instead -- tag bit 60: data-gradient tile, bit 61: weight-gradient workgroup)
#if defined(FCN_PROBE) && FCN_PROBE >= 3
#define BPROBE_DECL unsigned long long pb_[8]; int pbn_ = 0
#define BPROBE_STAMP() do { if (pbn_ < 7) pb_[pbn_++] = wall_clock64(); } while (0)
#define BPROBE_FLUSH(tag) PROBE_FLUSH(tag)
#else
#define BPROBE_DECL
#define BPROBE_STAMP()
#define BPROBE_FLUSH(tag)
#endif
#endif

#ifndef FCN_POOL_FUSED
#define FCN_POOL_FUSED 1     // 0 (tuning builds): conv3 writes y3 only and pool_nlc_kernel re-reads it
#endif
