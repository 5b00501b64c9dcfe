/*
 * fcn_hip.h -- C-ABI of libfcn_hip.so: the MI355X (gfx950) implementation of Frustum ConvNet's
 * per-frustum hot path.  Plain pointers + sizes + a HIP stream; no torch types, no hidden
 * allocation, no implicit synchronisation, no global state.  Every function enqueues work on
 * `stream` (a hipStream_t passed as void*) and returns 0 or a non-zero code (hipError_t value, or
 * FCN_E_* below); it is safe to call while the stream is being captured into a hipGraph.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *
 *   fcn_query_depth_point_f32   query_depth_point_forward(), PYBIND11 "forward"
 *                               ops/query_depth_point/query_depth_point_cuda.cpp:25-50 and the launcher +
 *                               kernel ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-86
 *   fcn_pn_compact / fcn_pn_forward / fcn_pn_backward
 *                               the torch ops behind PointNetModule.forward + the max over K in
 *                               PointNetFeat.forward: gather, centre subtract, 3 x [Conv2d 1x1, BatchNorm2d,
 *                               ReLU], (cnt>0) mask, torch.max(.,-1), one-hot concat
 *                               models/det_base.py:75-101,134-157 (+ autograd of the same)
 *   fcn_convnet_forward/backward ConvFeatNet.forward + cls_out/reg_out (cuDNN/ATen in the reference),
 *                               models/det_base.py:196-224,367-368 (+ autograd of the same)
 *   fcn_convnet_pack / _forward2 the same forward with the weight re-packing split off and per-feature-map start events
 *   fcn_det_loss_tail[_rows]    the ~150 torch ops of the train-loss tail, models/det_base.py:373-476
 *   fcn_adam_step_f32           optim.Adam.step() of the step loop, train/train_net_det.py:131-133,321-339
 *   fcn_sgd_step_f32            optim.SGD.step() (momentum) of the same loop's 'sgd' branch, train/train_net_det.py:325-327
 *   fcn_prepare_inputs          the per-sample numpy work of the data loader + collate,
 *                               datasets/provider_sample.py:137-262,270-327,396-397
 *   fcn_prepare_inputs_refine   the same for the refinement stage, datasets/provider_sample_refine.py:176-419
 *   fcn_prepare_inputs_sunrgbd  the same for the SUN-RGBD loader, datasets/provider_sample_sunrgbd.py:116-326
 *   fcn_det_iou_metrics         the IoU metrics of models/det_base.py:480-503 (D2H + boost clipping every step in the reference)
 *   fcn_box3d_iou_pair_f32      rbbox_iou_3d_pair, ops/pybind11/box_ops.h:173-260 (boost polygon clipping on the host)
 *   fcn_decode_detections       the numpy decode loop of train/test_net_det.py:254-293 + from_prediction_to_label_format
 *   fcn_rotate_nms_3d           rotate_nms_3d_cc, ops/pybind11/rbbox_iou.py:294-311 + nms_cpu.h:148-240
 *   fcn_stamp                   (measurement aid, no reference counterpart)
 *   fcn_stream_capture_id       (graph-capture bookkeeping of the Python layer, no reference counterpart)
 *
 * Buffers are caller-owned.  "ws" buffers are scratch the caller provides (sizes documented per call).
 */
#ifndef FCN_HIP_H
#define FCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCN_E_BADARG 10001   /* unsupported shape / null pointer */
#define FCN_E_LIMIT  10002   /* size beyond a documented limit   */

/* Matrix-core operand precision of the GEMM kernels (fcn_pn_desc.precision / fcn_cn_desc.precision).  Storage, BatchNorm
 * statistics and accumulation are fp32 in every mode.
 *   FCN_PREC_SPLIT  default, the parity mode: every fp32 operand is split into two 16-bit parts and a product is formed
 *                   from three 16-bit MFMAs (fp16 parts in the forward GEMMs, bf16 parts in the backward GEMMs) -- fp32-class
 *                   results (logits within 1e-4 of the fp32 reference) at 5.3x the fp32 matrix rate of gfx950.  Forward
 *                   operands (activations after BN+ReLU, weights) must stay below 65504 in magnitude.
 *   FCN_PREC_F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference mode for A/B comparisons.
 *   FCN_PREC_BF16   throughput mode of BASELINE config 2 (bf16 activations + GEMM inputs): single bf16 MFMA per product, fp32
 *                   accumulate, and the BIG intermediate tensors -- the PointNet's per-entry y2 / y3 / dy3 / dz2, the streams that
 *                   reach HBM -- STORED as bf16 in the first half of their (fp32-sized) buffers; BatchNorm sums, pooled features,
 *                   logits and all parameter gradients stay fp32 (logits within a few 1e-2 relative of the fp32 reference;
 *                   grouping indices unaffected).  The FCN's y / dz arenas (a few thousand rows per layer, L2-resident) stay
 *                   fp32 in this mode since round 6: bf16 arenas made every fcn_convnet_* launch 22-49 % slower and saved no
 *                   memory time (fcn_net.hip, CN_MM_OF).
 *   FCN_PREC_BF16_OPS  the same operands with fp32 storage everywhere (rounds 1-2's bf16 mode). */
#define FCN_PREC_SPLIT 0
#define FCN_PREC_F32   1
#define FCN_PREC_BF16  2
#define FCN_PREC_BF16_OPS 3

/* Version / build probe: returns 950 (the only arch this library is built for). */
int fcn_arch(void);

/* ---------------------------------------------------------------------------------------------
 * Sliding-frustum grouping.  Bit-exact restatement of query_depth_point_cuda_kernel.cu:16-65.
 *   pts_z : z of point 0 of sample 0; element stride pt_stride floats, sample stride pt_bstride floats
 *           ((B,3,N) layout: xyz+2*N, 1, 3*N;  (B,N,3) layout: xyz+2, 3, 3*N) -- never transposed on device
 *   ctr_z : same for the m window centres
 *   idx   : (b,m,nsample) int64, fully written (padding with first hit; zeros for empty windows)
 *   cnt   : (b,m) int32
 * ------------------------------------------------------------------------------------------- */
int fcn_query_depth_point_f32(const float *pts_z, int64_t pt_stride, int64_t pt_bstride,
                              const float *ctr_z, int64_t ct_stride, int64_t ct_bstride,
                              int b, int n, int m, float dis_z, int nsample,
                              int64_t *idx, int32_t *cnt, void *stream);

/* The same operator for ALL scales of one batch in ONE launch (the scales of PointNetFeat.forward share the point cloud and differ
 * in window centres, half height and nsample: models/det_base.py:126-157 calls query_depth_point once per scale).  Arrays of
 * nscale <= 8 entries; outputs exactly those of nscale calls of fcn_query_depth_point_f32. */
int fcn_query_depth_point_multi_f32(int nscale, const float *pts_z, int64_t pt_stride, int64_t pt_bstride,
                                    const float *const *ctr_z, const int64_t *ct_stride, const int64_t *ct_bstride,
                                    int b, int n, const int32_t *m, const float *dis_z, const int32_t *nsample,
                                    int64_t *const *idx, int32_t *const *cnt, void *stream);

/* ---------------------------------------------------------------------------------------------
 * One PointNet scale (PointNetModule + max over K), "entry space" dataflow: every distinct
 * (window, point) pair is evaluated once and carries its multiplicity as a weight.
 * Channel widths C1,C2,C3 must be multiples of 64; L <= 8192; K <= 1024.
 * ------------------------------------------------------------------------------------------- */
typedef struct fcn_pn_desc {
    int32_t B, N, L, K;          /* frustums, points per frustum, windows, nsample        */
    int32_t C1, C2, C3;          /* MLP widths                                              */
    int32_t nvec;                /* one-hot width appended after pooling (0..)              */
    int32_t training;            /* 1: batch statistics (+ running-stat update), 0: running */
    float   eps, momentum;       /* BatchNorm eps (1e-5) and momentum (0.1)                 */
    int32_t nlc;                 /* 0: feat/dfeat are (B, C3+nvec, L) as the reference returns them;
                                    1: position-major (B, L, C3), no one-hot rows (input of fcn_convnet_*) */
    int32_t precision;           /* FCN_PREC_* (0 = split 16-bit MFMA, fp32-class)                              */
    int32_t grouped;             /* 1: fcn_pn_group_compact prepared ws for this call (entry list, tile list, input moments,
                                    BN1 scale/shift + running statistics, zeroed BN sums): fcn_pn_forward skips its own
                                    BN1 finalisation                                                              */
} fcn_pn_desc;

/* Parameters of the three conv+BN pairs (reference state_dict order: conv{1,2,3}.0.weight,
 * conv{1,2,3}.1.{weight,bias,running_mean,running_var,num_batches_tracked}). */
typedef struct fcn_pn_params {
    const float *W[3];           /* (C1,3) (C2,C1) (C3,C2) row-major                        */
    const float *gamma[3], *beta[3];
    float *running_mean[3], *running_var[3];
    int64_t *num_batches_tracked[3];
} fcn_pn_params;

/* Scratch + saved-for-backward buffers of one scale.  cap = L*K rows per frustum.
 * All must stay alive from fcn_pn_forward until fcn_pn_backward has been enqueued. */
typedef struct fcn_pn_ws {
    int32_t *woff;               /* (B, L+1)   window row offsets, woff[b][L] = live rows    */
    float   *ent;                /* (B, cap, 4) (ux,uy,uz,w)                                 */
    int32_t *ewin;               /* (B, cap)    window of each row                           */
    int32_t *tiles;              /* 4 + B*ceil(cap/128): [0] = live 128-row tiles, [4+i] = b*tps + t */
    float   *y2;                 /* (B, cap, C2) conv2 output (pre-BN)                       */
    float   *y3;                 /* (B, cap, C3) conv3 output (pre-BN)                       */
    int32_t *amax;               /* (B, L, C3)  row of the pooled max, -1 = no gradient      */
    double  *stat;               /* 16 + fcn_stat_replicas() * (2*C2 + 2*C3) doubles: input moments, then the
                                    replicated sum / sumsq blocks of conv2 and conv3                        */
    float   *bn;                 /* 4*(C1+C2+C3) floats: per layer scale, shift, mean, rstd  */
    /* backward -- and, for gmax, the hand-over from a key-pooled TRAINING forward to its backward */
    float   *gmax;               /* (B, L, C3)  dfeat routed to the max rows (written by fcn_pn_backward*).  When the forward
                                    pooled through keys (training = 1, nlc = 1, pkey and ewin set) fcn_pn_forward ALSO writes
                                    it: the winners' pre-BN values, position-major, which the first backward kernel reads and
                                    then overwrites with the routed gradient.  Between such a forward and its backward the
                                    buffer is therefore LIVE: do not clear it, do not share it between scales or workspace
                                    sets in flight, and pass the same pointer to both calls.  A training key-pool forward
                                    with amax set and gmax NULL returns FCN_E_BADARG                                         */
    float   *dy3;                /* (B, cap, C3), or NULL: dy3 is not materialised -- conv3's weight-gradient GEMM rebuilds
                                    it from y3, ewin, amax, gmax and the BN3-backward sums while staging (bit-identical dW3;
                                    measured 0.7 % slower over the step, saves B*cap*C3 floats) */
    float   *dz2;                /* (B, cap, C2)                                             */
    double  *bstat;              /* fcn_stat_replicas() * (2*C3 + 2*C2 + 4*C1) doubles       */
    float   *coef;               /* 5*(C3+C2) floats                                         */
    float   *partial;            /* wgrad partials: nsplit * max(C3*C2, C2*C1) floats (fcn_pn_backward3: nsplit * (C3*C2 + C2*C1)) */
    int32_t  nsplit;             /* capacity of `partial` in splits: >= B*ceil(cap/128) (one per row tile) */
    double  *gmom;               /* (B,12) doubles, fcn_pn_group_compact only (may be NULL otherwise): per-frustum input moments +
                                    an arrival counter; zero it, and tiles[0..3], ONCE at allocation (the call leaves
                                    its counters at zero) */
    float   *wenc;               /* 2*(C2*C1 + C3*C2) floats, 16-byte aligned: the conv2 / conv3 weights split-encoded in MFMA
                                    operand order (forward + data-gradient images), rewritten by every forward
                                    (fcn_pn_pack_weights[_all]) and read by the forward and backward GEMMs                    */
    int32_t *flags;              /* 1 int32 of sticky FCN_FLAG_* bits, or NULL: zero it once, read it whenever convenient    */
    uint64_t *pkey;              /* (B, L, C3) max-pool keys, 16-byte aligned, or NULL: zero it ONCE at allocation (fcn_pn_forward
                                    leaves it zero).  With it (and nlc = 1) the max-pool is taken in conv3's epilogue instead of
                                    by a pass that re-reads y3 (and in eval mode y3 is not written at all)                     */
    int32_t  partial_both;       /* 1 or 2: `partial` holds nsplit * (C3*C2 + C2*C1) floats -- both weight gradients' split partials at
                                    once -- so their two fixed-order reduces (and the layer-1 finalisation) run as ONE launch at the
                                    tail of the chain instead of between its GEMMs; with 1 the one-stream backward (fcn_pn_backward,
                                    fcn_pn_backward2 with stream2 == NULL) also runs conv2's data gradient and both weight-gradient
                                    GEMMs as roles of one launch.  0: `partial` holds nsplit * max(C3*C2, C2*C1) floats, GEMM and
                                    reduce alternate.  Anything else: FCN_E_BADARG -- zero-initialise the struct.  Bit-identical
                                    gradients in all three. */
} fcn_pn_ws;

/* Sticky numeric flags (fcn_pn_ws.flags, fcn_cn_ws.flags): the kernels only ever OR bits in.
 *   FCN_FLAG_NONFINITE  a forward GEMM produced a non-finite output.  In FCN_PREC_SPLIT the forward operands are split into
 *                       fp16 parts, which overflow at |x| >= 65504 (the products turn into inf - inf = NaN and the next ReLU
 *                       would silently turn that into 0): results of this step are not to be trusted -- rerun in FCN_PREC_F32
 *                       or FCN_PREC_BF16, whose operands keep the fp32 exponent range. */
#define FCN_FLAG_NONFINITE 1

/* rows of one row tile (128): the caller sizes ws.tiles / ws.partial with it */
int fcn_pn_wgrad_rows(void);
/* copies of every BatchNorm sum slot (8): same-address fp64 atomics are served one at a time, so workgroups spread over
 * replicas and the consumers sum them in a fixed order; the caller sizes ws.stat / ws.bstat with it */
int fcn_stat_replicas(void);

/* idx/cnt -> entry list + live-tile list + weighted input moments (ws.woff, ws.ent, ws.ewin, ws.tiles, ws.stat[0..9]) */
int fcn_pn_compact(const fcn_pn_desc *d, const float *pc /*(B,3,N)*/, const float *ref /*(B,3,L)*/,
                   const int64_t *idx, const int32_t *cnt, const fcn_pn_ws *ws, void *stream);

/* Grouping + compaction of up to 8 scales of one batch in ONE launch, without the int64 idx: for scale s, window centres
 * ref[s] (B,3,L_s) and half height dis_z[s] produce cnt[s] (B,L_s) and ws[s]->{woff, ent, ewin, tiles, stat[0..9], bn (layer
 * 1)} exactly as fcn_query_depth_point_f32 + fcn_pn_compact + the BN1 finalisation of fcn_pn_forward would (moments up to
 * fp64 summation order, here fixed), zeroes the BN sum slots of ws[s]->stat and, in training mode, updates conv1's running
 * statistics.  All scales share pc (B,3,N), B, N, training, eps, momentum; N <= 65535.  Follow with fcn_pn_forward on
 * descriptors whose `grouped` field is 1. */
int fcn_pn_group_compact(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p, const float *pc,
                         const float *const *ref, const float *dis_z, const fcn_pn_ws *const *ws, int32_t *const *cnt,
                         void *stream);

/* The same front in two PHASES, so that a loader can prepare the NEXT batch while the current step still runs (the reference's
 * DataLoader workers prefetch batches the same way, datasets/provider_sample.py:291-327 behind train/train_net_det.py:114):
 *   phase 1  everything that depends on the batch alone -- hit lists, entry rows, window offsets, tile lists, input moments,
 *            zeroed BN sums -- into workspaces no launch in flight uses (two launches; p[s] must be valid but its weights are not read);
 *   phase 2  everything that depends on the WEIGHTS of the step that consumes the batch -- the split-encoded conv2 / conv3 images and
 *            the BN1 fold (scale / shift, running statistics) from the moments phase 1 left in ws[s]->stat -- one light launch,
 *            to be issued after the optimiser step, in front of fcn_pn_forward (descriptors with grouped = 1);
 *   phase 3  both at once = fcn_pn_group_compact.
 * Phases 1 + 2 leave every workspace bit-identical to phase 3 (tests/test_gpu_group_compact.py). */
int fcn_pn_group_compact2(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p, const float *pc,
                          const float *const *ref, const float *dis_z, const fcn_pn_ws *const *ws, int32_t *const *cnt,
                          int phase, void *stream);

/* Split-encodes the conv2 / conv3 weights of one scale into ws->wenc (one small launch).  fcn_pn_forward does it itself on
 * descriptors with grouped == 0; fcn_pn_group_compact does it for all its scales (fcn_pn_pack_weights_all, one launch). */
int fcn_pn_pack_weights(const fcn_pn_desc *d, const fcn_pn_params *p, const fcn_pn_ws *ws, void *stream);
int fcn_pn_pack_weights_all(int nscale, const fcn_pn_desc *const *d, const fcn_pn_params *const *p,
                            const fcn_pn_ws *const *ws, void *stream);

/* Whole forward of one scale after fcn_pn_compact: feat (B, C3+nvec, L), one_hot (B,nvec) or NULL.  In training mode its
 * last kernel also zeroes ws.bstat (when non-NULL) for the fcn_pn_backward that follows -- and, when the max-pool is taken
 * from keys (nlc = 1, ws.pkey and ws.ewin set), writes the winners' pre-BN values into ws.gmax, which the following
 * fcn_pn_backward* reads before overwriting it (see fcn_pn_ws.gmax; FCN_E_BADARG when ws.amax is set and ws.gmax is NULL). */
int fcn_pn_forward(const fcn_pn_desc *d, const fcn_pn_params *p, const int32_t *cnt,
                   const float *one_hot, const fcn_pn_ws *ws, float *feat, void *stream);

/* Backward: dfeat (B, C3+nvec, L) -> dW[3], dgamma[3], dbeta[3] (overwritten, not accumulated).  After a key-pooled training
 * forward (fcn_pn_forward above) it expects ws.gmax exactly as that forward left it.  dW[1] and dW[2] must be
 * 16-byte aligned (FCN_E_BADARG otherwise): the fixed-order sum of the split partials writes 16-byte vectors.  Size limits
 * (FCN_E_LIMIT): B * cap * max(C2, C3) < 2^31 elements (32-bit offsets) and B * cap < 2^24 entry rows (24-bit row multiplies). */
int fcn_pn_backward(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                    const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3], void *stream);

/* Same as fcn_pn_backward with the two weight-gradient GEMMs (+ their reduces) enqueued on a second stream beside the
 * data-gradient chain; events = 3 caller-owned hipEvent_t.  stream joins on the side stream before returning work. */
int fcn_pn_backward2(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                     const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                     void *stream, void *stream2, void *const *events);

/* Backward of the UN-POOLED module output -- PointNetModule.forward's (B, C3, L, nsample) return, models/det_base.py:62-103, whose
 * autograd the reference gets from torch.  dz3 (B, cap, C3) fp32: the gradient w.r.t. relu(bn3(y3)) of every ENTRY row (the K
 * slots of a window summed back onto its rows -- the first hit collects its K - ne + 1 duplicates --, the (cnt > 0) mask and the
 * ReLU mask applied); ws->bstat: replica 0 = sum dz3 [C3], sum dz3 * xhat3 [C3], everything else zero (the caller prepares both:
 * frustum_convnet_amd/pointnet_fused.py does it with device-side indexing).  One stream; needs ws->dy3; not in FCN_PREC_BF16. */
int fcn_pn_backward_dense(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dz3,
                          const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3], void *stream);

/* Three-way split: after the first data-gradient GEMM the chain continues on `stream` (dgrad of conv2, layer-1 finalisation),
 * conv3's weight gradient runs on stream2 and conv2's on stream3.  ws.partial must hold BOTH weight gradients' partials at once:
 * nsplit * (C3*C2 + C2*C1) floats.  events = 4 caller-owned hipEvent_t. */
int fcn_pn_backward3(const fcn_pn_desc *d, const fcn_pn_params *p, const float *dfeat,
                     const fcn_pn_ws *ws, float *dW[3], float *dgamma[3], float *dbeta[3],
                     void *stream, void *stream2, void *stream3, void *const *events);

/* Launches ONLY the conv GEMM of `layer` (2 or 3) on the state a previous fcn_pn_compact/fcn_pn_forward left in
 * ws: the unit the roofline figure in bench.py is measured on.  with_stats != 0 keeps the BN-statistics epilogue. */
int fcn_pn_conv_fwd(const fcn_pn_desc *d, const fcn_pn_params *p, const fcn_pn_ws *ws, int layer,
                    int with_stats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * ConvFeatNet + heads (models/det_base.py:163-224,250-251,365-368): 10 Conv1d + 3 ConvTranspose1d (each with
 * BatchNorm1d + ReLU) and the two k=1 heads, as implicit GEMMs over position-major (B*L, C) activations.
 * Layer order of every per-layer array (14 entries; entry 13 = heads, no BN):
 *   0 block1_conv1  1 block2_conv1  2 block2_conv2  3 block2_merge  4 block3_conv1  5 block3_conv2  6 block3_merge
 *   7 block4_conv1  8 block4_conv2  9 block4_merge  10 block2_deconv  11 block3_deconv  12 block4_deconv  13 heads
 * W[l]: the torch weight ((Cout,Cin,k) Conv1d / (Cin,Cout,k) ConvTranspose1d); W[13] = cat(cls_out.weight,
 * reg_out.weight) (2+reg_out, 768, 1), bias = cat of the two biases.
 * ------------------------------------------------------------------------------------------- */
#define FCN_CN_MAXLEV 5          /* pyramid levels: 4 (models/det_base.py) or 5 (models/det_base_sunrgbd.py)     */
#define FCN_CN_MAXLAYER 18       /* 4 * levels - 2 layers: 3*levels-2 convolutions, levels-1 deconvolutions, the heads */

typedef struct fcn_cn_desc {
    int32_t B;
    int32_t L[FCN_CN_MAXLEV];    /* positions of the pooled feature maps (L[1] is the output length); L[j+1] = conv(L[j], k3 s2 p1) */
    int32_t nvec;                /* one-hot width                                                               */
    int32_t reg_out;             /* regression head width (39 for KITTI, 67 for SUN-RGBD); 2 + reg_out <= 128   */
    int32_t training;
    float   eps, momentum;
    int32_t prepacked;           /* 1: fcn_convnet_pack already ran for these weights / one-hot (joined by the caller) */
    int32_t precision;           /* FCN_PREC_*                                                                  */
    int32_t nlev;                /* pyramid levels, 4 or 5 (0 = 4)                                              */
    int32_t c1;                  /* width of block1_conv1: 128 (det_base.py:168) or 64 (det_base_sunrgbd.py:178); 0 = 128 */
} fcn_cn_desc;

/* Layer order of the parameter arrays for n = nlev levels (torch module names of ConvFeatNet):
 *   0                      block1_conv1
 *   1 + 3*(j-2) + {0,1,2}  block{j}_conv1, block{j}_conv2, block{j}_merge      for j = 2..n
 *   3n-2 + (j-2)           block{j}_deconv                                      for j = 2..n
 *   4n-3                   heads: rows 0..1 cls_out, rows 2.. reg_out, over cat of the n-1 deconvolution outputs
 * (n = 4: 0 b1c1, 1-3 block2, 4-6 block3, 7-9 block4, 10-12 deconvs, 13 heads; n = 5: ... 10-12 block5, 13-16 deconvs, 17 heads) */
typedef struct fcn_cn_params {
    const float *W[FCN_CN_MAXLAYER];
    const float *gamma[FCN_CN_MAXLAYER], *beta[FCN_CN_MAXLAYER];
    float *running_mean[FCN_CN_MAXLAYER], *running_var[FCN_CN_MAXLAYER];
    int64_t *num_batches_tracked[FCN_CN_MAXLAYER];
    const float *bias;           /* (2 + reg_out) heads bias */
} fcn_cn_params;

/* Workspace; element counts come from fcn_convnet_sizes (out6: y/dz floats, packed-weight floats, bn floats,
 * stat/bstat doubles, coef floats, wgrad-partial floats).  Size limits (FCN_E_LIMIT from every fcn_convnet_* entry): B * L1 < 2^23
 * rows and every layer's B * L * C (and Cout * Ktot) < 2^30 elements -- the kernels address with 32-bit BYTE offsets. */
typedef struct fcn_cn_ws {
    float  *y, *dz, *wp, *bn;
    double *stat, *bstat;
    float  *coef, *partial;      /* coef: unused since BN backward is finalised by its consumers (kept in the layout) */
    float  *oh64;                /* B * 64 floats: the one-hot vector zero-padded to 64 channels */
    int32_t *flags;              /* 1 int32 of sticky FCN_FLAG_* bits, or NULL (see fcn_pn_ws.flags) */
} fcn_cn_ws;

int fcn_convnet_sizes(const fcn_cn_desc *d, int64_t *out6);
/* Weight re-packing (3.3 M floats, torch layouts -> (N, Ktot)) + one-hot padding of fcn_convnet_forward as a separate
 * launch, so a caller can overlap it with the PointNet scales on another stream (then set d->prepacked = 1).  The launch
 * also zeroes ws.stat and ws.bstat (training): fcn_convnet_forward / _backward enqueue no memset of their own. */
int fcn_convnet_pack(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws, const float *one_hot, void *stream);
/* feats[s], s < nlev: (B, L[s], C_s) with C = 128,128,256,512(,512) (fcn_pn_forward with nlc = 1); logits: (B*L[1], ld) rows
 * with ld = fcn_convnet_logits_ld(d) (64 when 2 + reg_out <= 64, else 128), columns 0..1 = cls_out, 2..2+reg_out = reg_out,
 * rest zero. */
int fcn_convnet_logits_ld(const fcn_cn_desc *d);
int fcn_convnet_forward(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws,
                        const float *const feats[FCN_CN_MAXLEV], const float *one_hot, float *logits, void *stream);
/* Same with nlev optional hipEvent_t: `stream` waits for feat_events[s] right before the first layer that reads feats[s], so
 * the FCN runs beside the PointNet scales that are still in flight (the widest map is needed by the last merge only). */
int fcn_convnet_forward2(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws,
                         const float *const feats[FCN_CN_MAXLEV], const float *one_hot, float *logits, void *stream,
                         void *const *feat_events);
int fcn_convnet_backward(const fcn_cn_desc *d, const fcn_cn_params *p, const fcn_cn_ws *ws,
                         const float *const feats[FCN_CN_MAXLEV], const float *one_hot, const float *dlogits,
                         float *const dfeats[FCN_CN_MAXLEV], float *const dW[FCN_CN_MAXLAYER],
                         float *const dgamma[FCN_CN_MAXLAYER], float *const dbeta[FCN_CN_MAXLAYER], float *dbias,
                         void *stream, void *stream2, void *const *events);
/* One launch per chain layer (the off-chain deconvolution steps ride along): its data-gradient tiles, its weight-gradient
 * row splits and the reduce of the previous layer's splits are workgroup roles of the same kernel.  stream2 / events: NULL,
 * or a second stream + nlev caller-owned events -- after the launch that completes dfeats[nlev-1] the chain continues on
 * stream2 (events[0] = fork, events[k] = dfeats[nlev-1-k] final for k = 1..nlev-2, events[nlev-1] = all outputs final), so
 * work queued on `stream` after the call waits for the widest map's gradient only. */

/* ---------------------------------------------------------------------------------------------
 * Fused train-loss tail of PointNetDet.forward (models/det_base.py:373-476; focal loss models/common.py:217-232,
 * huber + box corners models/model_util.py:9-19,48-72, encode/decode models/box_transform.py:5-65).
 *   cls_raw (B,2,L2), reg_raw (B,3+2*NB+4*NS,L2): raw head outputs;  cls_label (B,L2) int64 in {-1,0,1};
 *   center_ref2 (B,3,L2); box3d_center (B,3); box3d_heading (B,1); box3d_size (B,3); size_class (B,1) int64;
 *   mean_size (NS,3).  NB must be 12 and NS 3 (KITTI) or 10 (SUN-RGBD), else FCN_E_LIMIT.
 *   out16: total, cls, center, head_cls, head_res, size_cls, size_res, corners, cls_acc, head_acc, size_acc, nfg
 *   dcls / dreg: d(total)/d(cls_raw), d(total)/d(reg_raw), same layouts (NULL to skip).
 * ------------------------------------------------------------------------------------------- */
int fcn_det_loss_tail(const float *cls_raw, const float *reg_raw, const int64_t *cls_label,
                      const float *center_ref2, const float *box3d_center, const float *box3d_heading,
                      const float *box3d_size, const int64_t *size_class, const float *mean_size,
                      int B, int L2, int num_heading_bin, int num_size_cluster,
                      float w_box, float w_corner, float w_headreg, float w_sizereg,
                      float *out16, float *dcls, float *dreg, void *stream);
/* Same on the row-major logits of fcn_convnet_forward: logits (B*L2, ld), dlogits same shape (fully written); ld = 64 when
 * 2 + 3 + 2*NB + 4*NS <= 64 (KITTI), else 128 (SUN-RGBD). */
int fcn_det_loss_tail_rows(const float *logits, const int64_t *cls_label, const float *center_ref2,
                           const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                           const int64_t *size_class, const float *mean_size, int B, int L2,
                           int num_heading_bin, int num_size_cluster,
                           float w_box, float w_corner, float w_headreg, float w_sizereg,
                           float *out16, float *dlogits, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser step of the training loop (train/train_net_det.py:321-339 optim.Adam(lr, weight_decay); :131-133
 * optimizer.step()).  One streaming kernel over flat fp32 buffers of n elements (16-byte aligned), torch.optim.Adam
 * arithmetic (L2 weight decay, bias correction from the step counter).  Everything the host may change between steps
 * lives in device memory so the launch can sit inside a captured hipGraph:
 *   hyper6: lr, beta1, beta2, eps, weight_decay, grad_scale (grad is multiplied by grad_scale first: 1/world after a
 *           summing all-reduce);  step_slots: fcn_adam_step_slots(n) int64 counters, all equal (0 at start), every
 *           one advanced by the launch -- one per workgroup, so no workgroup waits on another. */
int64_t fcn_adam_step_slots(int64_t n);
int fcn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                      const float *hyper6, int64_t *step_slots, void *stream);

/* The 'sgd' branch of the same loop (train/train_net_det.py:325-327 optim.SGD(lr, momentum, weight_decay)): torch.optim.SGD
 * arithmetic (L2 weight decay, dampening 0, no Nesterov) over the same flat buffers; momentum_buf starts at zero.
 *   hyper4 (device): lr, momentum, weight_decay, grad_scale. */
int fcn_sgd_step_f32(float *param, const float *grad, float *momentum_buf, int64_t n, const float *hyper4, void *stream);

/* Same with a persistent scratch buffer (fcn_det_loss_tail_scratch_floats(B, L2) floats, zeroed ONCE by the caller, then
 * owned by one stream at a time): no memset node in front of the launch, the workgroup partials are summed in a fixed
 * order (the 16 scalars are reproducible bit for bit), and `total` (1 float, may be NULL) receives a copy of out16[0].
 * Alignment (all three loss-tail entry points): cls_label -- and, in the row-major forms, logits and dlogits -- must be 16-byte
 * aligned (the kernel reads two labels / four logits per load and stores four gradient values at once); FCN_E_BADARG otherwise. */
int fcn_det_loss_tail_scratch_floats(int B, int L2);
int fcn_det_loss_tail_rows2(const float *logits, const int64_t *cls_label, const float *center_ref2,
                            const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                            const int64_t *size_class, const float *mean_size, int B, int L2,
                            int num_heading_bin, int num_size_cluster,
                            float w_box, float w_corner, float w_headreg, float w_sizereg,
                            float *out16, float *dlogits, float *scratch, float *total, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rotated-box overlap after the heads.  The reference leaves the device for all of
 * it: numpy loops + boost::geometry polygon clipping on the host.
 * ------------------------------------------------------------------------------------------- */
/* rbbox_iou_3d_pair (ops/pybind11/box_ops.h:173-260, pybind "rbbox_iou_3d_pair"; call site models/det_base.py:495):
 * corners1/corners2 (n,8,3) in the corner order of get_box3d_corners_helper -> out2 (n,2) = [BEV IoU, 3-D IoU]. */
int fcn_box3d_iou_pair_f32(const float *corners1, const float *corners2, int n, float *out2, void *stream);
/* IoU training metrics of models/det_base.py:480-503 (which leaves the device for rbbox_iou_3d_pair on the host every step):
 * on the foreground rows (cls_label == 1) of the row-major logits (B*L2, ld), the box decoded with the arg-max heading bin /
 * size cluster against the label box.  out4 = mean BEV IoU, mean 3-D IoU, fraction with 3-D IoU >= iou_thresh
 * (cfg.IOU_THRESH), foreground count.  scratch8: 8 floats, zeroed ONCE by the caller (the launch leaves them zero). */
int fcn_det_iou_metrics(const float *logits, int ld, const int64_t *cls_label, const float *center_ref2,
                        const float *box3d_center, const float *box3d_heading, const float *box3d_size,
                        const float *mean_size, int B, int L2, int num_heading_bin, int num_size_cluster,
                        float iou_thresh, float *scratch8, float *out4, void *stream);
/* The per-frustum decode loop of train/test_net_det.py:254-293 on the row-major logits (B*L2, ld) of
 * fcn_convnet_forward (cols 0..1 cls, 2.. reg): method 1 ('nms'): every position with p_bg < p_fg, or the arg-max of p_fg
 * when a frustum has none; method 0 ('top'): the arg-max only.  Arg-max heading bin / size cluster decode, centre =
 * offset + center_ref2 (B,3,L2), from_prediction_to_label_format (datasets/provider_sample.py:375-387) with rot_angle (B),
 * ref_center (B,3) or NULL (zeros), rgb_prob (B) or NULL (ones).
 *   dets (B*L2, 8) = tx, ty, tz, l, w, h, ry, score (every row written); valid (B*L2) = selected and h,w,l >= 0.01. */
int fcn_decode_detections(const float *logits, int ld, const float *center_ref2, const float *mean_size,
                          const float *rot_angle, const float *ref_center, const float *rgb_prob, int B, int L2,
                          int num_heading_bin, int num_size_cluster, int method, float *dets, int32_t *valid,
                          void *stream);
/* rotate_nms_3d_cc (ops/pybind11/rbbox_iou.py:294-311) -> rotate_non_max_suppression_3d_cpu (ops/pybind11/nms_cpu.h:148-240)
 * for num_groups independent detection sets at once (one per (frame, class), train/test_net_det.py:126-152): dets rows are
 * organised in units of rows_per_unit consecutive rows (one frustum each), unit_group (num_units) in [0, num_groups) assigns
 * units to groups, valid (rows) or NULL marks the candidate rows.  Greedy in descending score, suppress when the rotated
 * 3-D IoU >= thresh.  keep (num_groups, top_k): row indices in keep order; keep_cnt (num_groups): count, or -1 when a group
 * holds more than 4096 candidates (FCN_E_LIMIT condition reported per group, the other groups are still processed). */
int fcn_rotate_nms_3d(const float *dets, const int32_t *valid, const int32_t *unit_group, int num_units, int rows_per_unit,
                      int num_groups, float thresh, int top_k, int32_t *keep, int32_t *keep_cnt, void *stream);

/* Measurement aid: stores the device's constant-rate wall clock (100 MHz ticks) into *slot, in stream order. */
int fcn_stamp(uint64_t *slot, void *stream);

/* Which hipGraph capture `stream` is part of, asked of the HIP runtime this library launches on: *id = 0 when the stream is not
 * capturing, the runtime's capture id + 1 otherwise.  Non-zero return: the query failed (hipError_t) or the capture was
 * invalidated (FCN_E_BADARG) -- treat the id as unknown.  No reference counterpart (the reference captures no graphs). */
int fcn_stream_capture_id(void *stream, uint64_t *id);

/* 16 hex characters: sha256 over the kernel sources, headers and compile flags this library was built from
 * (frustum_convnet_amd/build.py source_hash()).  The library travels prebuilt; __graft_entry__.smoke() and bench.py compare
 * it with the tree they run from, so a stale binary cannot pass for the committed sources. */
const char *fcn_build_hash(void);


/* ---------------------------------------------------------------------------------------------
 * On-device construction of one training batch from raw frustum records: replaces the
 * per-sample numpy work of datasets/provider_sample.py::ProviderDataset.__getitem__ (:137-262; generate_ref :291-327,
 * generate_labels :270-289, centre-view helpers :329-372) and torch's default_collate (:396-397).
 *   raw_pts (sum n_b, pt_stride) float32 records in rect camera coordinates, pt_off (B+1) int64 row offsets,
 *   raw_seg (sum n_b) int64 or NULL, choice (B,N) int32 resample indices (the reference's np.random.choice),
 *   frustum_angle (B), box2d (B,4), P (B,3,4), box3d_corners (B,8,3), heading (B), size (B,3) (l,w,h), coin (B)
 *   (flip when > 0.5), normal (B) (depth-shift draw): all fp64 like the pickled records.
 * Outputs (caller-owned): point_cloud (B,3,N), center_ref[s] (B,3,L[s]), cls_label (B,L[1]) int64 in {-1,0,1} (NULL to
 * skip), box3d_center (B,3), box3d_heading (B,1), box3d_size (B,3), rot_angle (B,1), seg_label (B,N) int64 or NULL. */
typedef struct fcn_inp_desc {
    int32_t B, N, pt_stride;     /* frustums, points per frustum after resampling, floats per raw record (>= 3) */
    int32_t L[4];                /* centres per stride: len(arange(0, max_depth, stride[s])) */
    double  stride[4], max_depth;
    int32_t random_flip, random_shift;
} fcn_inp_desc;
int fcn_prepare_inputs(const fcn_inp_desc *d, const float *raw_pts, const int64_t *pt_off, const int64_t *raw_seg,
                       const int32_t *choice, const double *frustum_angle, const double *box2d, const double *P,
                       const double *box3d_corners, const double *heading, const double *size,
                       const double *coin, const double *normal, float *point_cloud, float *const center_ref[4],
                       int64_t *cls_label, float *box3d_center, float *box3d_heading, float *box3d_size,
                       float *rot_angle, int64_t *seg_label, void *stream);

/* SUN-RGBD variant (cfgs/det_sample_sunrgbd.yaml; datasets/provider_sample_sunrgbd.py::ProviderDataset.__getitem__ :116-263,
 * generate_ref :283-326, project_image_to_upright_camera :28-59): five strides; the window centres go through the camera
 * matrix K (B,3,3) and the tilt rotation Rtilt (B,3,3) -- camera (x,y,z) -> (x,z,-y) -> Rtilt . -> (X,-Z,Y) -- instead of
 * the KITTI projection; random_shift also draws a height shift, hshift (B) = the np.random.random() value in [0,1), applied
 * as hshift*0.4-0.2 to the points' y and the box centre.  Everything else as fcn_prepare_inputs. */
typedef struct fcn_inp5_desc {
    int32_t B, N, pt_stride;
    int32_t L[5];
    double  stride[5], max_depth;
    int32_t random_flip, random_shift;
} fcn_inp5_desc;
int fcn_prepare_inputs_sunrgbd(const fcn_inp5_desc *d, const float *raw_pts, const int64_t *pt_off, const int64_t *raw_seg,
                               const int32_t *choice, const double *frustum_angle, const double *box2d, const double *K,
                               const double *Rtilt, const double *box3d_corners, const double *heading, const double *size,
                               const double *coin, const double *normal, const double *hshift, float *point_cloud,
                               float *const center_ref[5], int64_t *cls_label, float *box3d_center, float *box3d_heading,
                               float *box3d_size, float *rot_angle, int64_t *seg_label, void *stream);

/* Refinement-stage variant (cfgs/refine_car.yaml; datasets/provider_sample_refine.py::ProviderDataset.__getitem__ :176-315,
 * generate_ref :336-386, generate_labels :317-334, collate_fn :388-419): the sample is normalised to the first-stage
 * prediction (pred_corners (B,8,3), pred_angle (B), pred_size (B,3) = l,w,h, fp64 like the pickled records): points and label
 * box are translated to the predicted centre and rotated by its heading; the window centres span the predicted box's own
 * depth extent -- a different count per sample; every center_ref[s] (B,3,Lpad[s]) and cls_label (B,Lpad[1]) is padded by
 * repeating the last real entry, as the reference's collate_fn does (Lpad = the batch maxima, computed by the caller as
 * max_b len(np.arange(-w_b/2, w_b/2, stride[s]))); lens (B,4) receives the per-sample counts.  box3d_corners == NULL:
 * inference records (no labels; cls_label / box3d_* must be NULL).  Outputs rot_angle (B,1) = pred_angle and
 * ref_center (B,3) = predicted centre are what from_prediction_to_label_format needs to undo the normalisation. */
typedef struct fcn_inp_refine_desc {
    int32_t B, N, pt_stride;
    int32_t Lpad[4];
    double  stride[4];
    int32_t random_flip, random_shift;
} fcn_inp_refine_desc;
int fcn_prepare_inputs_refine(const fcn_inp_refine_desc *d, const float *raw_pts, const int64_t *pt_off,
                              const int32_t *choice, const double *pred_corners, const double *pred_angle,
                              const double *pred_size, const double *box3d_corners, const double *heading,
                              const double *size, const double *coin, const double *normal, float *point_cloud,
                              float *const center_ref[4], int64_t *cls_label, float *box3d_center, float *box3d_heading,
                              float *box3d_size, float *rot_angle, float *ref_center, int32_t *lens, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FCN_HIP_H */
