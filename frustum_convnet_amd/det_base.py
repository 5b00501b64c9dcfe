"""Drop-in for the reference's models/det_base.py: PointNetModule, PointNetFeat, ConvFeatNet, PointNetDet.

Same constructors, forward signatures, return structures and state_dict keys (154 entries:
feat_net.pointnet{1-4}.conv{1-3}.{0.weight,1.*}, conv_net.block*.{0.weight,1.*}, cls_out.*, reg_out.*), so
checkpoints written by the reference's train/train_net_det.py:384-398 load unchanged.

What runs where (all of it hand-written HIP behind the C-ABI of include/fcn_hip.h; the nn.Conv*/BatchNorm* children only
HOLD the parameters so that state_dicts stay interchangeable):
  * grouping + shared MLP + max-pool of every scale: csrc/grouping.hip, pointnet_fwd.hip, pointnet_bwd.hip
    (PointNetModule.forward_pooled / launch_pooled);
  * ConvFeatNet + heads: csrc/fcn_net.hip (implicit-GEMM forward + backward) -- `fused_fcn`;
  * train-loss tail (8 losses, 3 accuracies, 3 IoU metrics, d total / d logits): csrc/loss_tail.hip + csrc/box_iou.hip,
    one launch each (the metrics on a side stream), no host synchronisation (the reference syncs at det_base.py:70, :414 and :495 every step) -- `fused_loss`;
  * eval decode: torch ops on the device (box_ops.py).
`fused_fcn = False` / `fused_loss = False` are EXPLICIT opt-ins to the nn.Conv1d / torch-op formulations (GPU libraries,
kept for A/B parity tests); nothing falls back to them silently.
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import cfg
from .dataset_info import DATASET_INFO
from .common import Conv1d, Conv2d, DeConv1d, init_params, softmax_focal_loss_ignore, get_accuracy, masked_mean, bn_momentum
from .query_depth_point import QueryDepthPoint
from .pointnet_fused import WorkspacePool, pointnet_pooled, launch_pooled, attach_pooled
from . import box_ops
from . import _native


class PointNetModule(nn.Module):
    """Single-scale PointNet (reference: models/det_base.py:35-103)."""

    def __init__(self, Infea, mlp, dist, nsample, use_xyz=True, use_feature=True):
        super(PointNetModule, self).__init__()
        self.dist = dist
        self.nsample = nsample
        self.use_xyz = use_xyz
        self.use_feature = Infea > 0
        if self.use_feature or not use_xyz:
            raise NotImplementedError(
                "only the xyz-only path (Infea == 0) is implemented: every shipped cfg has "
                "WITH_EXTRA_FEAT False and train_net_det.py:296 makes Infea <= 0 otherwise")
        self.query_depth_point = QueryDepthPoint(dist, nsample)
        self.conv1 = Conv2d(Infea + 3, mlp[0], 1)
        self.conv2 = Conv2d(mlp[0], mlp[1], 1)
        self.conv3 = Conv2d(mlp[1], mlp[2], 1)
        init_params([self.conv1[0], self.conv2[0], self.conv3[0]], 'kaiming_normal')
        init_params([self.conv1[1], self.conv2[1], self.conv3[1]], 1)
        self._pool = WorkspacePool()

    def _param_pack(self):
        convs = (self.conv1, self.conv2, self.conv3)
        params = []
        for c in convs:
            params += [c[0].weight, c[1].weight, c[1].bias]
        bufs = ([c[1].running_mean for c in convs], [c[1].running_var for c in convs],
                [c[1].num_batches_tracked for c in convs])
        return params, bufs

    def forward_pooled(self, pc, new_pc, one_hot_vec=None, nlc=False):
        """Fused fast path: max over K of the masked features, one-hot appended -> (B, C3+nvec, L).
        Equals torch.max(self.forward(pc, None, new_pc), -1)[0] (+ the concat of PointNetFeat.forward).
        nlc=True: position-major (B, L, C3) without the one-hot rows (what the fused ConvFeatNet consumes)."""
        params, bufs = self._param_pack()
        bn = self.conv1[1]
        feat, _, _ = pointnet_pooled(self._pool, self.dist, self.nsample, self.training, bn.eps,
                                     bn_momentum(bn),
                                     pc.contiguous(), new_pc.contiguous(), one_hot_vec, bufs, params, nlc=nlc)
        return feat

    def launch_pooled(self, pc, new_pc, one_hot_vec=None, nlc=False):
        """Enqueue this scale's forward kernels now, create the autograd node later with attach_pooled()."""
        params, bufs = self._param_pack()
        bn = self.conv1[1]
        return launch_pooled(self._pool, self.dist, self.nsample, self.training, bn.eps,
                             bn_momentum(bn),
                             pc.contiguous(), new_pc.contiguous(), one_hot_vec, bufs, params, nlc=nlc)

    def prepare_pooled(self, pc, new_pc, one_hot_vec=None, nlc=False):
        """Workspace / descriptor of this scale for the fused front (pointnet_fused.group_compact + launch_prepared)."""
        from .pointnet_fused import prepare_pooled
        params, bufs = self._param_pack()
        bn = self.conv1[1]
        return prepare_pooled(self._pool, self.dist, self.nsample, self.training, bn.eps, bn_momentum(bn),
                              pc.contiguous(), new_pc.contiguous(), one_hot_vec, bufs, params, nlc=nlc)

    def attach_pooled(self, handle):
        feat, _, _ = attach_pooled(self._pool, handle)
        return feat

    def front_signature(self, nlc=False):
        """Everything a prepared handle (prepare_pooled) FREEZES besides the input tensors: the configuration tuple (training,
        need_grad, BatchNorm eps / momentum, layout), the operand precision and where the parameters and BatchNorm buffers live.
        A prefetched front is only consumed by a forward that would have prepared the same handle."""
        from .pointnet_fused import _cfg_tuple
        from . import precision as _precision
        params, bufs = self._param_pack()
        bn = self.conv1[1]
        cfgt = _cfg_tuple(self.dist, self.nsample, self.training, bn.eps, bn_momentum(bn), params, nlc)
        return (cfgt, _precision.code(), tuple(t.data_ptr() for t in params), tuple(t.data_ptr() for b in bufs for t in b))

    def forward(self, pc, feat, new_pc=None):
        """Reference-shaped output (B, C3, L, nsample), masked (models/det_base.py:62-103), expanded from the fused path's per-entry
        activations with device-side indexing.  With gradients enabled (training mode, a parameter that requires them) the
        return carries its graph, as the reference's does: the backward sums the slots back onto the entry rows and runs the
        HIP backward chain (pointnet_fused.dense_pointnet).  Training inside PointNetDet goes through forward_pooled, which never
        materialises the K slots."""
        from .pointnet_fused import dense_from_entries, dense_pointnet
        params, bufs = self._param_pack()
        bn = self.conv1[1]
        if pc.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("PointNetModule.forward: no gradient w.r.t. the point cloud (the reference never asks for one either: "
                               "its inputs do not require grad)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            if not self.training:
                # (the reference returns a differentiable tensor here too -- fine-tuning under frozen BatchNorm; the HIP backward
                # chain differentiates batch-statistics BatchNorm only, so this raises instead of handing back a tensor without a graph)
                raise NotImplementedError(
                    "PointNetModule.forward in eval mode with gradients enabled and trainable parameters: the backward of "
                    "running-statistics BatchNorm is not implemented; wrap the call in torch.no_grad() (inference) or call "
                    ".train() (training)")
            return dense_pointnet(self._pool, self.dist, self.nsample, True, bn.eps, bn_momentum(bn),
                                  pc.contiguous(), new_pc.contiguous(), bufs, params)
        with torch.no_grad():
            return dense_from_entries(self._pool, self.dist, self.nsample, self.training, bn.eps,
                                      bn_momentum(bn),
                                      pc.contiguous(), new_pc.contiguous(), bufs, params)


class PointNetFeat(nn.Module):
    """Four scales (reference: models/det_base.py:107-159).  SCALES = (mlp widths, nsample) per scale; the 5-scale
    SUN-RGBD variant (det_base_sunrgbd.py) overrides it."""

    SCALES = (([64, 64, 128], 32), ([64, 64, 128], 64), ([128, 128, 256], 64), ([256, 256, 512], 128))

    def __init__(self, input_channel=3, num_vec=0):
        super(PointNetFeat, self).__init__()
        self.num_vec = num_vec
        u = cfg.DATA.HEIGHT_HALF
        assert len(u) == len(self.SCALES)
        self.num_scales = len(self.SCALES)
        for i, (mlp, nsample) in enumerate(self.SCALES):        # children pointnet1..pointnetN, as the reference names them
            setattr(self, "pointnet%d" % (i + 1),
                    PointNetModule(input_channel - 3, list(mlp), u[i], nsample, use_xyz=True, use_feature=True))
        self.concurrent_scales = True
        self.fused_front = os.environ.get("FCN_FUSED_FRONT", "1") != "0"
        self._stream_cache = {}
        self._prefetched = None     # prefetch(): the next batch's front, phase 1 (key, prepared handles, event, capture id)
        # the widest scale is the long pole of the backward: its weight-gradient GEMMs run on a second stream beside its
        # data-gradient chain (set_wgrad_streams changes the choice).  Other stream topologies -- every scale forked, no join in
        # front of the FCN, the FCN backward continuing on a second stream, other capture orders -- measured slower on ROCm 7.2 /
        # MI355X in rounds 3-5 (EXPERIMENTS.md) and are not options of this layer any more.
        self.set_wgrad_streams((self.num_scales - 1,))
        # guest scale -> host scale: the guest's backward is enqueued on the host's stream, behind the host's chain (one parallel
        # branch less in a captured step's backward: share_backward_stream).  Applied only while `share_active` is set -- by
        # PointNetDet.forward on its fused-FCN path, where ONE launch sequence produces every scale's gradient.
        self.bwd_share = dict(self.BWD_SHARE.get(self.num_scales, ()))
        self.share_active = False
        self._step_id = 0

    def set_wgrad_streams(self, scales, three=False):
        """Which scales (0-based) run their weight-gradient GEMMs on a second stream (fcn_pn_backward2; `three`: conv2's on a third,
        fcn_pn_backward3).  Bit-identical gradients in every setting (tests/test_gpu_model.py)."""
        for k, net in enumerate(self.nets):
            net._pool.side_wgrad = k in tuple(scales)
            net._pool.side_three = bool(three) and k in tuple(scales)

    # measured on MI355X / ROCm 7.2 (EXPERIMENTS 6.12): 4 scales: 1 -> 2 (car -1.5 %, people -1.0 %, refine -3.3 %, bf16 mode -2.3 %);
    # 5 scales (SUN-RGBD): 0 -> 2 and 1 -> 3 (-1.9 %)
    BWD_SHARE = {4: ((1, 2),), 5: ((0, 2), (1, 3))}

    def share_backward_stream(self, scale, host):
        """The backward of `scale` (0-based) is enqueued on the stream of scale `host`, BEHIND that scale's backward chain, instead of
        on its own stream: one parallel branch less in a captured step's backward -- ROCm 7.2's graph executor deals the branches onto
        four internal streams, and the PointNet backward of a 4-scale model has five (the scales + the widest scale's weight-gradient
        stream): the one left over starts ~200 us late (EXPERIMENTS 4.2 / 6.12).  host = None restores the scale's own stream.
        Requires scale < host < num_scales - 1: autograd runs the scales' nodes in reverse creation order (forward() creates them
        0 .. n-1), so the host's node has ALREADY run when the guest's kernels are enqueued on its stream.  Eager launches get event
        edges both ways (correct for any graph).  Under hipGraph capture ROCm 7.2 cannot take an edge between two forked streams, so
        stream order is the only edge: valid when every scale's incoming gradient is final before ANY scale's node runs -- the fused
        ConvFeatNet backward (one call produces them all); PointNetDet sets `share_active` on that path only, and a guest whose host
        has not been differentiated for the same forward, or is not part of the same capture, stays on its own stream
        (pointnet_fused._PointNetPooled.backward).  Same kernels on the same data: bit-identical gradients."""
        ns = self.num_scales
        if host is None:
            self.bwd_share.pop(scale, None)
            return
        if not (0 <= scale < host < ns - 1):
            raise ValueError("share_backward_stream: need 0 <= scale < host < %d (got scale %d, host %d)" % (ns - 1, scale, host))
        self.bwd_share[scale] = host

    @property
    def nets(self):
        return tuple(getattr(self, "pointnet%d" % (i + 1)) for i in range(self.num_scales))

    def _front_key(self, point_cloud, sample_pc, one_hot_vec, nlc, training):
        # the inputs as prefetch() saw them (storage, shape, version counter; a non-contiguous tensor is keyed as passed) and
        # what the prepared handles froze: configuration, precision, parameter / buffer storage of every scale
        ts = [point_cloud] + list(sample_pc) + ([] if one_hot_vec is None else [one_hot_vec])
        return (tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version) for t in ts), bool(nlc), bool(training),
                torch.is_grad_enabled(), tuple(net.front_signature(nlc) for net in self.nets))

    def prefetch(self, point_cloud, sample_pc, one_hot_vec=None, nlc=False, before=None):
        """Phase 1 of the fused front (grouping, entry rows, tile lists, input moments: functions of the batch alone) for the
        batch the NEXT forward() will see, on a side stream forked from the current one -- as a data loader prefetches
        (datasets/provider_sample.py:291-327 run by DataLoader workers ahead of train/train_net_det.py:114).  The next forward
        then starts with the light weight-dependent launch (phase 2) instead of the whole front.  The tensors must be passed
        to that forward unmodified (same storage, no in-place write in between); anything else discards the prefetch.
        join_prefetch() makes the current stream wait for the branch (PointNetDet.backward does: inside a captured step
        the branch has to end in the same capture).  before: called on the branch's stream in FRONT of the front's launches --
        the place of a launch that PRODUCES the batch on the device (inputs.InputBuilder.launch into these very tensors)."""
        if not (self.fused_front and self.concurrent_scales and point_cloud.is_cuda):
            return False
        self.drop_prefetch()
        from .pointnet_fused import group_compact
        dev = point_cloud.device
        nets = self.nets
        cur = torch.cuda.current_stream(dev)
        key = "pf" + str(dev)
        if key not in self._stream_cache:
            self._stream_cache[key] = (torch.cuda.Stream(device=dev), torch.cuda.Event(enable_timing=False),
                                       torch.cuda.Event(enable_timing=False))
        side, ev_in, ev_out = self._stream_cache[key]
        prepared = [nets[s].prepare_pooled(point_cloud, sample_pc[s], one_hot_vec, nlc) for s in range(self.num_scales)]
        ev_in.record(cur)
        side.wait_event(ev_in)
        with torch.cuda.stream(side):
            if before is not None:
                before()
            group_compact(prepared, point_cloud, phase=1)
            ev_out.record(side)
        for t in [point_cloud] + list(sample_pc):
            t.record_stream(side)
        # cap: the hipGraph capture these launches belong to (0: none -- they really ran)
        self._prefetched = {"key": self._front_key(point_cloud, sample_pc, one_hot_vec, nlc, self.training), "handles": prepared,
                            "event": ev_out, "dev": dev, "cap": _native.capture_id(dev)}
        return True

    def _prefetch_is_foreign(self):
        """A prefetch made while a hipGraph was being captured exists only INSIDE that capture: its launches have not run (and
        never will if the capture was aborted) and its event belongs to the capture.  Outside it -- an eager forward after the
        capture, another capture -- the entry is discarded: neither consumed nor waited for.  (Under graph REPLAY the prefetched
        front reads the input buffers the capture saw: refill BOTH steps' buffers before a replay.)"""
        pf = self._prefetched
        return pf is not None and pf["cap"] != _native.capture_id(pf["dev"])

    def adopt_prefetch(self):
        """For a step that is SEVERAL graphs replayed in a fixed order: a prefetch made inside an earlier graph of the cycle is
        handed to the capture now in progress.  The caller guarantees that every replay of this graph follows a replay of that one
        on the same stream (so the branch has run; no event crosses the two captures)."""
        pf = self._prefetched
        if pf is not None:
            pf["cap"] = _native.capture_id(pf["dev"])
            pf["event"] = None

    def join_prefetch(self):
        """The current stream waits for the prefetch branch (no-op without one, or when the branch belongs to another capture)."""
        pf = self._prefetched
        if pf is not None and pf["event"] is not None:
            if not self._prefetch_is_foreign():
                torch.cuda.current_stream(pf["dev"]).wait_event(pf["event"])
            pf["event"] = None

    def drop_prefetch(self):
        """Forgets a prefetched front nobody consumed (its workspaces go back to the pools)."""
        if self._prefetched is None:
            return
        self.join_prefetch()
        for net, h in zip(self.nets, self._prefetched["handles"]):
            net._pool.release(h["ws"])
        self._prefetched = None

    def forward(self, point_cloud, sample_pc, feat=None, one_hot_vec=None, nlc=False, join=True):
        """join=False (fused FCN path): the caller's stream is NOT made to wait for the scales; self.done_events holds one
        event per scale for the consumer to wait on (fcn_convnet_forward2 does, map by map)."""
        if one_hot_vec is not None:
            assert self.num_vec == one_hot_vec.shape[1]
        nets = self.nets
        ns = self.num_scales
        self.done_events = None
        if not (self.concurrent_scales and point_cloud.is_cuda) or os.environ.get("FCN_SERIAL", "0") == "1":
            self.drop_prefetch()
            for net in nets:
                net._pool.bwd_stream = net._pool.bwd_host_pool = None
            return tuple(net.forward_pooled(point_cloud, ref, one_hot_vec, nlc) for net, ref in zip(nets, sample_pc))
        # The scales are independent until the FCN: all but the last run on HIP streams forked from the current one and
        # the widest (the last scale, the long pole) on the current stream itself, captured as parallel branches of the step's
        # hipGraph, so one scale's tail (a few workgroups left on 256 CUs) overlaps the others' work.
        #  * Forks are flat: ROCm 7.2 stream capture crashes on a fork from an already-forked stream, and the widest scale's
        #    backward forks a second stream for its weight-gradient GEMMs -- hence it stays on the current stream.
        #  * Launch order is heaviest first (4, 3, 1, 2); node creation order is 1, 2, 3, 4: autograd replays each scale's
        #    backward on the stream its node was created under, in REVERSE creation order -- widest first again.
        #    launch_pooled() / attach_pooled() separate the two orders.
        dev = point_cloud.device
        cur = torch.cuda.current_stream(dev)
        streams = self._streams(dev)
        fork = self._fork_event(dev)
        # fused front: grouping + compaction + BN1 of all scales in ONE launch on the caller's stream, in front of the fork
        # (fcn_pn_group_compact2); fused_front = False keeps the API-form grouping per scale (int64 idx, 5 nodes each)
        prepared = None
        if self._prefetched is not None and (self._prefetch_is_foreign() or
                                             self._prefetched["key"] != self._front_key(point_cloud, sample_pc, one_hot_vec, nlc,
                                                                                        self.training)):
            self.drop_prefetch()
        if self.fused_front:
            from .pointnet_fused import group_compact, launch_prepared
            if self._prefetched is not None:
                # the batch-only part ran ahead (prefetch): only the weight-dependent launch is left on the chain
                self.join_prefetch()
                prepared = self._prefetched["handles"]
                self._prefetched = None
                group_compact(prepared, point_cloud, phase=2)
            else:
                prepared = [nets[s].prepare_pooled(point_cloud, sample_pc[s], one_hot_vec, nlc) for s in range(ns)]
                group_compact(prepared, point_cloud)
        fork.record(cur)
        sts = [streams[i] for i in range(ns - 1)] + [cur]
        self._step_id += 1
        for k in range(ns):                  # where each scale's backward will be enqueued (read by its autograd node)
            host = self.bwd_share.get(k) if (self.share_active and self.training) else None
            pool = nets[k]._pool
            pool.fwd_step = self._step_id
            pool.bwd_stream = None if host is None else streams[host]
            pool.bwd_host_pool = None if host is None else nets[host]._pool
        handles = [None] * ns
        for s in (ns - 1,) + tuple(range(ns - 2, 1, -1)) + (0, 1):           # heaviest first; 4 scales: (3, 2, 0, 1)
            if sts[s] is not cur:
                sts[s].wait_event(fork)
            with torch.cuda.stream(sts[s]):
                if prepared is not None:
                    handles[s] = launch_prepared(prepared[s])
                else:
                    handles[s] = nets[s].launch_pooled(point_cloud, sample_pc[s], one_hot_vec, nlc)
        outs = [None] * ns
        done = self._done_events(dev)
        for s in range(ns):
            with torch.cuda.stream(sts[s]):
                outs[s] = nets[s].attach_pooled(handles[s])
                done[s].record(sts[s])
        # (the join is always made: consuming the maps one by one without it -- the FCN's first layers beside the widest scale --
        # measured slower; fcn_convnet_forward2 still accepts per-map events)
        for s in range(ns):
            cur.wait_event(done[s])
            outs[s].record_stream(cur)
        return tuple(outs)

    def _done_events(self, device):
        key = "done" + str(device)
        if key not in self._stream_cache:
            self._stream_cache[key] = [torch.cuda.Event(enable_timing=False) for _ in range(self.num_scales)]
        return self._stream_cache[key]

    def _fork_event(self, device):
        key = "ev" + str(device)
        if key not in self._stream_cache:
            self._stream_cache[key] = torch.cuda.Event(enable_timing=False)
        return self._stream_cache[key]

    def _streams(self, device):
        key = str(device)
        if key not in self._stream_cache:
            self._stream_cache[key] = [torch.cuda.Stream(device=device) for _ in range(self.num_scales)]
        return self._stream_cache[key]


class ConvFeatNet(nn.Module):
    """Conv1d FCN over the stacked frustum feature maps (reference: models/det_base.py:163-224).  LEVELS / WIDTHS describe
    the pyramid: block1_conv1 has WIDTHS[0] channels, block{j}_* WIDTHS[j-1] (= the width of pooled map j), and
    block{j}_deconv upsamples level j by 2^(j-2) to 256 channels; det_base_sunrgbd.py overrides them (5 levels, 64-wide block1)."""

    LEVELS = 4
    WIDTHS = (128, 128, 256, 512)
    DECONV_FIRST = 2          # registration order of the deconvolutions (= state_dict / optimizer parameter order):
                              # block2..block4 here (det_base.py:181-183), block5..block2 in det_base_sunrgbd.py:196-199

    def __init__(self, i_c=128, num_vec=3):
        super(ConvFeatNet, self).__init__()
        w = self.WIDTHS
        assert len(w) == self.LEVELS
        self.block1_conv1 = Conv1d(i_c + num_vec, w[0], 3, 1, 1)
        for j in range(2, self.LEVELS + 1):
            setattr(self, "block%d_conv1" % j, Conv1d(w[j - 2], w[j - 1], 3, 2, 1))
            setattr(self, "block%d_conv2" % j, Conv1d(w[j - 1], w[j - 1], 3, 1, 1))
            setattr(self, "block%d_merge" % j, Conv1d(w[j - 1] + w[j - 1] + num_vec, w[j - 1], 1, 1))
        levels = range(2, self.LEVELS + 1)
        for j in (levels if self.DECONV_FIRST == 2 else reversed(levels)):
            k = 1 << (j - 2)
            setattr(self, "block%d_deconv" % j, DeConv1d(w[j - 1], 256, k, k, 0))
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
                nn.init.kaiming_normal_(m.weight.data, mode='fan_in')
                if m.bias is not None:
                    m.bias.data.zero_()
            if isinstance(m, nn.BatchNorm1d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, *xs):
        assert len(xs) == self.LEVELS
        x = self.block1_conv1(xs[0])
        ups = []
        for j in range(2, self.LEVELS + 1):
            x = getattr(self, "block%d_conv1" % j)(x)
            x = getattr(self, "block%d_conv2" % j)(x)
            x = getattr(self, "block%d_merge" % j)(torch.cat([x, xs[j - 1]], 1))
            ups.append(getattr(self, "block%d_deconv" % j)(x))
        n = ups[0].shape[-1]
        return torch.cat([ups[0]] + [u[:, :, :n] for u in ups[1:]], 1)


class _PendingPointNetBackward:
    """Phase 2 of a split backward (PointNetDet.take_split)."""

    def __init__(self, model, feats, leaves):
        self.model, self.feats, self.leaves = model, feats, leaves

    def backward(self, scales=None):
        """scales: None = every scale not differentiated yet; or an iterable of 0-based scale indices -- the step loop may
        differentiate the wide scales first and start the all-reduce of their gradients (FlatTrainState.allreduce_scales_async)
        while the narrow ones run.  The object is finished when every scale has been differentiated."""
        if any(l.grad is None for l in self.leaves):
            raise RuntimeError("phase 2 of the split backward before phase 1: differentiate the loss first")
        if not hasattr(self, "_todo"):
            self._todo = set(range(len(self.feats)))
        ks = sorted(self._todo if scales is None else set(scales))
        if not set(ks) <= self._todo:
            raise RuntimeError("scales %s were differentiated already (left: %s)" % (sorted(set(ks) - self._todo), sorted(self._todo)))
        if ks:
            # (widest first, as the one-call form does: autograd runs the nodes in reverse creation order within a call)
            torch.autograd.backward([self.feats[k] for k in ks], [self.leaves[k].grad for k in ks])
        self._todo -= set(ks)
        if self._todo:
            return
        self.model._join_side()
        if self.model._pending_split is self:
            self.model._pending_split = None
        self.feats = self.leaves = None


class PointNetDet(nn.Module):
    """Whole pipeline (reference: models/det_base.py:228-525)."""

    FEAT_NET = PointNetFeat
    CONV_NET = ConvFeatNet

    def __init__(self, input_channel=3, num_vec=0, num_classes=2):
        super(PointNetDet, self).__init__()
        dataset_name = cfg.DATA.DATASET_NAME
        assert dataset_name in DATASET_INFO
        self.category_info = DATASET_INFO[dataset_name]
        self.num_size_cluster = len(self.category_info.CLASSES)
        self.mean_size_array = self.category_info.MEAN_SIZE_ARRAY
        self.feat_net = self.FEAT_NET(input_channel, num_vec)
        self.conv_net = self.CONV_NET(128, num_vec)
        self.num_scales = self.feat_net.num_scales
        assert self.conv_net.LEVELS == self.num_scales
        self.num_classes = num_classes
        self.num_bins = cfg.DATA.NUM_HEADING_BIN
        output_size = 3 + self.num_bins * 2 + self.num_size_cluster * 4
        head_in = 256 * (self.num_scales - 1)              # 768 (det_base.py:252-253) / 1024 (det_base_sunrgbd.py:278-279)
        self.reg_out = nn.Conv1d(head_in, output_size, 1)
        self.cls_out = nn.Conv1d(head_in, 2, 1)
        self.relu = nn.ReLU(True)
        nn.init.kaiming_uniform_(self.cls_out.weight, mode='fan_in')
        nn.init.kaiming_uniform_(self.reg_out.weight, mode='fan_in')
        self.cls_out.bias.data.zero_()
        self.reg_out.bias.data.zero_()
        # strict=True reproduces the reference's per-step host checks (fg assert) and IoU metrics
        # through `iou_fn(corners_pred, corners_gt) -> (n,2)` when one is supplied.
        # constant table kept on the module's device (not in the state_dict: the reference rebuilds it from
        # numpy every forward, det_base.py:357 -- an H2D copy that cannot be captured in a hipGraph)
        self.register_buffer("_mean_size", torch.tensor(self.mean_size_array, dtype=torch.float32),
                             persistent=False)
        self.strict = False
        self.iou_fn = None
        # fused_loss: the whole train-loss tail (values + d total/d logits) in one HIP launch (csrc/loss_tail.hip);
        # False keeps the mask-weighted torch formulation below (needed for the optional IoU metrics).
        self.fused_loss = True
        # fused_fcn: ConvFeatNet + heads as hand-written implicit-GEMM HIP kernels over position-major activations
        # (csrc/fcn_net.hip); False runs the nn.Conv1d / BatchNorm1d modules through MIOpen.
        self.fused_fcn = True
        # split_backward: forward() cuts the autograd graph at the pooled feature maps; backward_split(loss, between) then
        # runs [loss, heads, ConvFeatNet] first and the four PointNet scales second, calling `between()` in the middle --
        # where the data-parallel step starts the all-reduce of the FCN gradients (2.9 M of the 3.3 M parameters) so that it
        # overlaps the PointNet backward (train/train_net_det.py:126-128 + nn.DataParallel's reduce, :308-309)
        self.split_backward = False
        self._split = None
        self._pending_split = None
        self._zero_cache = {}
        self._loss_scratch = None
        from .fcn_fused import CnPool
        self._cn_pool = CnPool()
        from .loss_fused import IouMetrics
        self._iou_metrics = IouMetrics()
        # defer_metrics_join: the IoU-metrics branch (side stream) is joined by backward() / backward_split() / the pending
        # phase-2 object instead of right behind the loss tail, so that the first backward launch does not wait for it (nothing
        # in the backward reads it).  OPT-IN: a training loop that sets it promises to differentiate through one of those three
        # (a plain loss.backward() would leave the branch unjoined until the next forward) and to read the metrics after it.
        self.defer_metrics_join = False
        self.last_logits = None
        self.last_logits64 = None
        self.last_num_fg = None

    def detect(self, data_dicts, unit_group=None, num_groups=None, method=None, thresh=None, top_k=300):
        """Inference tail on the device (train/test_net_det.py:193-293 + :126-152): eval forward, decode into label-format
        boxes, rotated 3-D NMS per (frame, class) group.  data_dicts may carry 'rot_angle' (B,1), 'ref_center' (B,3) and
        'rgb_prob' (B,1) as the reference's test loader does (:201-214 defaults: zeros / ones).  unit_group (B,) int32 maps
        each frustum to its (frame, class) group (default: every frustum its own group).
        Returns dets (B*L2, 8) [tx,ty,tz,l,w,h,ry,score], valid (B*L2,), keep (G, top_k), keep_cnt (G,) -- device tensors;
        with method 'top' (cfg.TEST.METHOD default) there is one candidate per frustum and no suppression is run."""
        from . import detect as fdet
        if self.training:
            raise RuntimeError("detect() runs in eval mode")
        method = cfg.TEST.METHOD if method is None else method
        thresh = cfg.TEST.THRESH if thresh is None else thresh
        dd = {k: v for k, v in data_dicts.items() if k not in ('box3d_center', 'rot_angle', 'ref_center', 'rgb_prob')}
        with torch.no_grad():
            self.forward(dd)
        if self.last_logits64 is None:
            raise RuntimeError("detect() needs the fused ConvFeatNet (fused_fcn = True)")
        refs2 = data_dicts['center_ref2']
        B, _, L2 = refs2.shape
        dev = refs2.device
        rot = data_dicts.get('rot_angle')
        rot = torch.zeros(B, device=dev) if rot is None else rot.to(dev)
        dets, valid = fdet.decode_detections(self.last_logits64, refs2, self._mean_size, rot, data_dicts.get('ref_center'),
                                             data_dicts.get('rgb_prob'), self.num_bins, self.num_size_cluster, method)
        if unit_group is None:
            unit_group = torch.arange(B, dtype=torch.int32, device=dev)
            num_groups = B
        elif num_groups is None:
            num_groups = int(unit_group.max().item()) + 1
        keep, cnt = fdet.rotate_nms_3d(dets, valid, unit_group, L2, num_groups, thresh if method == 'nms' else 2.0, top_k)
        return dets, valid, keep, cnt

    def numeric_flags(self, device=None):
        """OR of the sticky FCN_FLAG_* bits the forward GEMMs of this model have raised so far, as a device int32 tensor (no
        host synchronisation; .item() it whenever convenient -- every N steps, at a checkpoint).  Bit FCN_FLAG_NONFINITE (1): a
        forward GEMM produced a non-finite output -- in the default split precision the fp16 operand parts overflow at
        |x| >= 65504, and the following ReLU would turn the resulting NaN into a silent zero."""
        pools = [n._pool for n in self.feat_net.nets] + [self._cn_pool]
        ts = [t for pl in pools for k, t in pl._flags.items() if device is None or k == str(device)]
        if not ts:
            return torch.zeros(1, dtype=torch.int32, device=device if device is not None else self.reg_out.weight.device)
        out = ts[0].clone()
        for t in ts[1:]:
            out |= t.to(out.device)       # (pools used on several devices: combine on the first one)
        return out

    def check_numerics(self):
        """Raises FloatingPointError when a numeric flag is up (synchronises); clears the flags."""
        f = int(self.numeric_flags().item())
        for pl in [n._pool for n in self.feat_net.nets] + [self._cn_pool]:
            for t in pl._flags.values():
                t.zero_()
        if f & 1:
            raise FloatingPointError("frustum_convnet_amd: a forward GEMM produced a non-finite value (fp16 operand overflow in "
                                     "split precision, |x| >= 65504, or a genuine overflow); rerun with precision 'f32' or 'bf16'")
        return f

    def prefetch(self, data_dicts, before=None):
        """Starts the batch-only part of the NEXT forward's front (sliding-frustum grouping, entry rows, tile lists, input
        moments of all scales) on a side stream, beside whatever the current stream does next -- call it between
        `model(data)` and `model.backward(loss)` with the batch the next `model(...)` call will get (the same tensors,
        unmodified), as a DataLoader worker prepares the next batch while the step runs (train/train_net_det.py:114).
        backward() / backward_split() join the branch; the next forward then begins with one light launch (weight images +
        BN1 fold) instead of the front.  Returns False when the prefetch does not apply (CPU tensors, module path).
        before: a callable run on the branch in front of the front -- e.g. the launch that BUILDS that batch on the device from
        resident raw records (inputs.InputBuilder.launch(t, out=data_dicts)); `model.next_batch_build` hands one to the prefetch
        that forward() starts for `model.next_batch`."""
        pc = data_dicts.get('point_cloud')
        if pc is None or not pc.is_cuda or not self.fused_fcn or data_dicts.get('one_hot') is None:
            return False
        refs = [data_dicts.get('center_ref%d' % i) for i in range(1, self.num_scales + 1)]
        xyz = pc[:, :3, :].contiguous()
        if before is not None and xyz.data_ptr() != pc.data_ptr():
            raise RuntimeError("prefetch(before=...): the batch is produced on the prefetch branch, so its point cloud must be the "
                               "(B, 3, N) tensor the front reads (a copy of the coordinate rows would be taken before it exists)")
        self._pf_xyz = ((pc.data_ptr(), tuple(pc.shape), pc._version), xyz)
        return self.feat_net.prefetch(xyz, refs, data_dicts.get('one_hot'), nlc=True, before=before)

    def _join_side(self):
        self._iou_metrics.join()
        self.feat_net.join_prefetch()

    def backward(self, loss):
        """loss.backward() seeded with a cached unit gradient (loss_fused.unit_grad): two tiny kernels (ones fill, multiply by
        one) less between the loss tail and the first backward GEMM.  Same gradients."""
        from .loss_fused import unit_grad
        if self._split is not None:
            return self.backward_split(loss)
        loss.backward(gradient=unit_grad(loss.device))
        self._join_side()

    def take_split(self):
        """Hands over phase 2 of a split backward (the PointNet scales' part of the graph) as an object with .backward();
        the caller differentiates the loss itself first (phase 1).  Until the returned object's backward() has run, the
        model counts as holding a half-finished backward: the next forward and FlatTrainState's optimiser step raise."""
        if self._split is None:
            raise RuntimeError("take_split needs a training forward with model.split_backward = True")
        pending = _PendingPointNetBackward(self, *self._split)
        self._split = None
        self._pending_split = pending
        return pending

    def backward_pending(self):
        """True while phase 2 of a split backward has not run: the PointNet gradients in the flat buffer are the previous
        step's (the HIP backward overwrites, nothing zeroes them)."""
        return self._split is not None or self._pending_split is not None

    def backward_split(self, loss, between=None):
        """loss.backward() in two phases (needs split_backward = True at forward time): phase 1 differentiates the loss
        tail, heads and ConvFeatNet (their parameter gradients are final when it returns), `between()` runs, phase 2
        differentiates the PointNet scales.  Numerically identical to loss.backward()."""
        from .loss_fused import unit_grad
        pending = self.take_split()
        loss.backward(gradient=unit_grad(loss.device))
        if between is not None:
            between()
        pending.backward()
        self._join_side()

    def _slice_output(self, output):
        nb, ns = self.num_bins, self.num_size_cluster
        center = output[:, 0:3]
        heading_scores = output[:, 3:3 + nb]
        heading_res_norm = output[:, 3 + nb:3 + 2 * nb]
        size_scores = output[:, 3 + 2 * nb:3 + 2 * nb + ns]
        size_res_norm = output[:, 3 + 2 * nb + ns:].reshape(output.shape[0], ns, 3)
        return center, heading_scores, heading_res_norm, size_scores, size_res_norm

    def _zero_scalar(self, like):
        """A cached constant 0 (the IoU metric placeholders): no fill kernel per step."""
        key = (str(like.device), like.dtype)
        z = self._zero_cache.get(key)
        if z is None:
            z = torch.zeros((), dtype=like.dtype, device=like.device)
            self._zero_cache[key] = z
        return z

    def forward(self, data_dicts):
        point_cloud = data_dicts.get('point_cloud')
        one_hot_vec = data_dicts.get('one_hot')
        cls_label = data_dicts.get('cls_label')
        size_class_label = data_dicts.get('size_class')
        center_label = data_dicts.get('box3d_center')
        heading_label = data_dicts.get('box3d_heading')
        size_label = data_dicts.get('box3d_size')
        refs = [data_dicts.get('center_ref%d' % i) for i in range(1, self.num_scales + 1)]

        batch_size = point_cloud.shape[0]
        pf = getattr(self, "_pf_xyz", None)
        self._pf_xyz = None
        if pf is not None and pf[0] == (point_cloud.data_ptr(), tuple(point_cloud.shape), point_cloud._version):
            xyz = pf[1]                  # the coordinate slice prefetch() already handed to the front
        else:
            xyz = point_cloud[:, :3, :].contiguous()
        mean_size_array = self._mean_size.to(device=point_cloud.device, dtype=point_cloud.dtype)

        logits64 = None
        self._iou_metrics.join()          # (a deferred join nobody made: the branch must not outlive its step)
        if self.backward_pending():
            # (a plain loss.backward() after a split forward differentiates only the loss tail, heads and ConvFeatNet)
            self._split = self._pending_split = None
            raise RuntimeError("PointNetDet: the previous forward ran with split_backward = True and its backward was never "
                               "finished (use model.backward(loss) / backward_split(loss) / take_split().backward()): the "
                               "PointNet gradients would silently be the previous step's")
        if self.fused_fcn and not point_cloud.is_cuda:
            raise RuntimeError("frustum_convnet_amd: the hot path runs on an MI355X only (got a %s tensor); there is no "
                               "CPU fallback" % point_cloud.device)
        if self.fused_fcn and one_hot_vec is None:
            raise NotImplementedError("the fused ConvFeatNet reads the one-hot class vector (data_dicts['one_hot']); set "
                                      "model.fused_fcn = False explicitly to run the nn.Conv1d modules instead")
        if self.fused_fcn:
            from .fcn_fused import convnet_fused, convnet_prepack
            # the FCN's weight re-packing depends on the weights only: it runs on a side stream beside the PointNet scales, forked
            # in FRONT of the grouping front (behind it, or captured behind its launches, measured slower)
            dev = point_cloud.device
            pre = convnet_prepack(self._cn_pool, self.conv_net, self.cls_out, self.reg_out, batch_size,
                                  [r.shape[2] for r in refs], one_hot_vec, dev)
            self.feat_net.share_active = True          # (every scale's gradient comes out of ONE fcn_convnet_backward call)
            feats = self.feat_net(xyz, refs, None, one_hot_vec, nlc=True, join=False)
            if self.split_backward and self.training and torch.is_grad_enabled():
                # two-phase backward (backward_split): the FCN sees detached leaves, so loss.backward() stops at the pooled
                # feature maps and the PointNet scales are differentiated by a second call
                leaves = tuple(f.detach().requires_grad_(True) for f in feats)
                self._split = (feats, leaves)
                feats = leaves
            nxt = getattr(self, "next_batch", None)
            if nxt is not None and self.training and torch.is_grad_enabled():
                # the caller announced the NEXT batch (model.next_batch = data_dicts): start its batch-only front here, on a side
                # branch beside the latency-bound ConvFeatNet forward whose launches leave most CUs idle (instead of a
                # model.prefetch() call between forward and backward, which lands beside the first backward launches)
                self.next_batch = None
                build, self.next_batch_build = getattr(self, "next_batch_build", None), None
                self.prefetch(nxt, before=build)
            logits64 = convnet_fused(self._cn_pool, self.conv_net, self.cls_out, self.reg_out, feats, one_hot_vec, pre,
                                     self.feat_net.done_events)
            lv = logits64.view(batch_size, refs[1].shape[2], logits64.shape[1])
            nreg = self.reg_out.weight.shape[0]
            cls_raw = lv[:, :, 0:2].permute(0, 2, 1)
            reg_raw = lv[:, :, 2:2 + nreg].permute(0, 2, 1)
        else:
            self.feat_net.share_active = False
            x = self.conv_net(*self.feat_net(xyz, refs, None, one_hot_vec))
            cls_raw = self.cls_out(x)
            reg_raw = self.reg_out(x)
        self.last_logits = (cls_raw, reg_raw)
        self.last_logits64 = logits64

        num_out = reg_raw.shape[2]
        fused_tail = (center_label is not None and self.fused_loss and cls_raw.is_cuda and self.iou_fn is None
                      and not self.strict and self.num_bins == 12 and self.num_size_cluster in (3, 10))
        if not fused_tail:       # the fused loss tail reads the raw logits itself: none of these copies / softmax
            cls_scores = cls_raw.permute(0, 2, 1).reshape(-1, 2)
            outputs = reg_raw.permute(0, 2, 1).reshape(-1, reg_raw.shape[1])
            center_ref2 = refs[1].permute(0, 2, 1).reshape(-1, 3)
            cls_probs = F.softmax(cls_scores, -1)

        if center_label is None:
            assert not self.training, 'Please provide labels for training.'
            center_boxnet, heading_scores, heading_res_norm, size_scores, size_res_norm = self._slice_output(outputs)
            heading_probs = F.softmax(heading_scores, -1)
            size_probs = F.softmax(size_scores, -1)
            heading_pred_label = torch.argmax(heading_probs, -1)
            size_pred_label = torch.argmax(size_probs, -1)
            center_preds = center_boxnet + center_ref2
            heading_preds = box_ops.angle_decode(heading_res_norm, heading_pred_label, num_bins=self.num_bins)
            size_preds = box_ops.size_decode(size_res_norm, mean_size_array, size_pred_label)
            return (cls_probs.view(batch_size, -1, 2), center_preds.view(batch_size, -1, 3),
                    heading_preds.view(batch_size, -1), size_preds.view(batch_size, -1, 3),
                    heading_probs.view(batch_size, -1, self.num_bins),
                    size_probs.view(batch_size, -1, self.num_size_cluster))

        if fused_tail:
            from .loss_fused import det_loss_tail, det_loss_tail_rows, loss_scratch
            Lw = cfg.LOSS
            wts = (Lw.BOX_LOSS_WEIGHT, Lw.CORNER_LOSS_WEIGHT, Lw.HEAD_REG_WEIGHT, Lw.SIZE_REG_WEIGHT)
            if logits64 is not None:
                key = (batch_size, num_out, str(logits64.device))
                if self._loss_scratch is None or self._loss_scratch[0] != key:
                    self._loss_scratch = (key, loss_scratch(batch_size, num_out, logits64.device))
                # IoU metrics (models/det_base.py:480-503) on a side stream beside the loss tail: validate()'s best-checkpoint
                # criterion (train/train_net_det.py:203,365-382) works without the reference's per-step D2H + boost clipping
                ious = self._iou_metrics(logits64, batch_size, num_out, cls_label, refs[1], center_label, heading_label,
                                         size_label, mean_size_array, self.num_bins, self.num_size_cluster, cfg.IOU_THRESH)
                losses, (a_cls, a_head, a_size), nfg = det_loss_tail_rows(
                    logits64, batch_size, num_out, cls_label, refs[1], center_label, heading_label, size_label,
                    size_class_label, mean_size_array, self.num_bins, self.num_size_cluster, wts,
                    self._loss_scratch[1])
                if not (self.defer_metrics_join and self.training and torch.is_grad_enabled()):
                    self._iou_metrics.join()
                iou2, iou3, iout = ious[0], ious[1], ious[2]
            else:
                losses, (a_cls, a_head, a_size), nfg = det_loss_tail(
                    cls_raw, reg_raw, cls_label, refs[1], center_label, heading_label, size_label, size_class_label,
                    mean_size_array, self.num_bins, self.num_size_cluster, wts)
                iou2 = iou3 = iout = self._zero_scalar(a_cls)       # (planar A/B path: no IoU metrics)
            self.last_num_fg = nfg      # device scalar: 0 means the batch had no foreground row (the reference asserts there)
            metrics = {'cls_acc': a_cls, 'head_acc': a_head, 'size_acc': a_size, 'IoU_2D': iou2, 'IoU_3D': iou3,
                       'IoU_' + str(cfg.IOU_THRESH): iout}
            return losses, metrics

        # ---- training / validation branch: every loss is a mean over the foreground rows
        # (cls_label == 1); written as mask-weighted sums over all B*L2 rows -> no nonzero(), no sync.
        lab = cls_label.view(-1)
        fg = (lab == 1).to(outputs.dtype)
        nfg = fg.sum()
        if self.strict:
            assert int(nfg.item()) != 0
        center_boxnet, heading_scores, heading_res_norm, size_scores, size_res_norm = self._slice_output(outputs)
        heading_probs = F.softmax(heading_scores, -1)
        size_probs = F.softmax(size_scores, -1)
        cls_loss = softmax_focal_loss_ignore(cls_probs, lab, ignore_idx=-1)

        center_lab = center_label.unsqueeze(1).expand(-1, num_out, -1).reshape(-1, 3)
        heading_lab = heading_label.expand(-1, num_out).reshape(-1)
        size_lab = size_label.unsqueeze(1).expand(-1, num_out, -1).reshape(-1, 3)
        size_cls_lab = size_class_label.expand(-1, num_out).reshape(-1)

        center_gt_offsets = box_ops.center_encode(center_lab, center_ref2)
        heading_cls_lab, heading_res_lab = box_ops.angle_encode(heading_lab, num_bins=self.num_bins)
        size_res_lab = box_ops.size_encode(size_lab, mean_size_array, size_cls_lab)

        mm = lambda v: masked_mean(v, fg, nfg)
        center_dist = torch.norm(center_gt_offsets - center_boxnet, 2, dim=-1)
        center_loss = mm(box_ops.huber_elem(center_dist, 3.0))
        heading_class_loss = mm(F.cross_entropy(heading_scores, heading_cls_lab, reduction='none'))
        hres_sel = torch.gather(heading_res_norm, 1, heading_cls_lab.view(-1, 1)).squeeze(1)
        heading_res_norm_loss = mm(box_ops.huber_elem(hres_sel - heading_res_lab, 1.0))
        size_class_loss = mm(F.cross_entropy(size_scores, size_cls_lab, reduction='none'))
        sres_sel = torch.gather(size_res_norm, 1, size_cls_lab.view(-1, 1, 1).expand(-1, 1, 3)).squeeze(1)
        size_res_norm_loss = mm(box_ops.huber_elem(torch.norm(size_res_lab - sres_sel, 2, dim=-1), 1.0))

        center_preds = box_ops.center_decode(center_ref2, center_boxnet)
        heading = box_ops.angle_decode(heading_res_norm, heading_cls_lab, num_bins=self.num_bins)
        size = box_ops.size_decode(size_res_norm, mean_size_array, size_cls_lab)
        corners_gt = box_ops.get_box3d_corners_helper(center_lab, heading_lab, size_lab)
        corners_gt_flip = box_ops.get_box3d_corners_helper(center_lab, heading_lab + np.pi, size_lab)
        corners_pred = box_ops.get_box3d_corners_helper(center_preds, heading, size)
        corners_dist = torch.min(torch.norm(corners_pred - corners_gt, 2, dim=-1).mean(-1),
                                 torch.norm(corners_pred - corners_gt_flip, 2, dim=-1).mean(-1))
        corners_loss = mm(box_ops.huber_elem(corners_dist, 1.0))

        L = cfg.LOSS
        loss = cls_loss + L.BOX_LOSS_WEIGHT * (
            center_loss + heading_class_loss + size_class_loss + L.HEAD_REG_WEIGHT * heading_res_norm_loss +
            L.SIZE_REG_WEIGHT * size_res_norm_loss + L.CORNER_LOSS_WEIGHT * corners_loss)

        with torch.no_grad():
            cls_prec = get_accuracy(cls_probs, lab, ignore=-1)
            heading_prec = get_accuracy(heading_probs, heading_cls_lab, mask=fg)
            size_prec = get_accuracy(size_probs, size_cls_lab, mask=fg)
            zero = torch.zeros((), dtype=cls_prec.dtype, device=cls_prec.device)
            iou2d_mean = iou3d_mean = iou3d_gt_mean = zero
            if self.iou_fn is not None:
                keep = fg.bool()
                hp = box_ops.angle_decode(heading_res_norm, torch.argmax(heading_probs, -1), num_bins=self.num_bins)
                sp = box_ops.size_decode(size_res_norm, mean_size_array, torch.argmax(size_probs, -1))
                cp = box_ops.get_box3d_corners_helper(center_preds, hp, sp)
                ov = self.iou_fn(cp[keep].detach().cpu().numpy(), corners_gt[keep].detach().cpu().numpy())
                iou2d_mean = torch.tensor(ov[:, 0].mean()).type_as(cls_prec)
                iou3d_mean = torch.tensor(ov[:, 1].mean()).type_as(cls_prec)
                iou3d_gt_mean = torch.tensor((ov[:, 1] >= cfg.IOU_THRESH).mean()).type_as(cls_prec)

        losses = {'total_loss': loss, 'cls_loss': cls_loss, 'center_loss': center_loss,
                  'head_cls_loss': heading_class_loss, 'head_res_loss': heading_res_norm_loss,
                  'size_cls_loss': size_class_loss, 'size_res_loss': size_res_norm_loss,
                  'corners_loss': corners_loss}
        metrics = {'cls_acc': cls_prec, 'head_acc': heading_prec, 'size_acc': size_prec,
                   'IoU_2D': iou2d_mean, 'IoU_3D': iou3d_mean, 'IoU_' + str(cfg.IOU_THRESH): iou3d_gt_mean}
        return losses, metrics
