"""Host side of the fused PointNet scale: workspaces + autograd binding over the C-ABI.

One call = one scale of models/det_base.py's PointNetFeat: grouping -> (gather, centre, 3 x [1x1 conv, BN,
ReLU], mask, max over K, one-hot concat) -> pooled (B, C3+nvec, L), forward and backward, all in HIP
(frustum_convnet_amd/csrc).  Nothing here computes on the host or falls back to torch ops.
"""
import ctypes
import os

import torch

from . import _native
from ._native import PnDesc, PnParams, PnWs
from .query_depth_point import query_depth_point
from . import precision as _precision


def _mid_launch(B, L, K, C3):
    """Merged middle launch of a scale's backward (fcn_pn_ws.partial_both)?  It pays where the scale's chain is LATENCY-bound: two
    launches less on a chain of small kernels.  Measured on MI355X (round 4, one-box A/Bs): car -0.4 ... -0.9 % per step with it on
    the two narrow scales (C3 = 128, 8 960 slots per frustum; ROCm's graph executor queues one of them behind the widest scale's
    data-gradient branch, so it runs at the tail of the phase, alone: 100 -> 89 us), refine -1.5 ... -3 % (L = 20 ... 3: every chain
    is launch latency); people +0.6 % SLOWER (its narrow scales hold 22 400 slots per frustum: the merged launch's roles share one
    register / LDS budget and last as long as the three they replace), SUN-RGBD +/-0; on a 128-wide scale nothing, for the widest
    scale instead of its second stream +5 %.  FCN_PN_MID: auto (default: C3 <= 128 and at most 300 k slots) | 1 (every one-stream
    scale) | 0; FCN_PN_MID_L=140,280: by window count (tuning)."""
    mode = os.environ.get("FCN_PN_MID", "auto")
    if str(L) in os.environ.get("FCN_PN_MID_L", "").split(","):
        return True
    return mode == "1" or (mode == "auto" and C3 <= 128 and B * L * K <= 300000)


class Workspace:
    """Caller-owned scratch of one scale (fcn_pn_ws).  Persistent across steps, recycled through the
    owning module's free list so that two forwards in flight (before their backwards) never share one."""

    def __init__(self, B, N, L, K, C1, C2, C3, device, need_grad, flags=None):
        cap = L * K
        f32, i32, f64 = torch.float32, torch.int32, torch.float64
        dev = device
        self.key = (B, N, L, K, C1, C2, C3, bool(need_grad))
        self.woff = torch.empty((B, L + 1), dtype=i32, device=dev)
        self.ent = torch.empty((B, cap, 4), dtype=f32, device=dev)
        self.ewin = torch.empty((B, cap), dtype=i32, device=dev)
        rows = _native.lib().fcn_pn_wgrad_rows()
        rep = _native.lib().fcn_stat_replicas() if hasattr(_native.lib(), "fcn_stat_replicas") else 1
        ntile_max = B * ((cap + rows - 1) // rows)
        self.tiles = torch.zeros((4 + ntile_max,), dtype=i32, device=dev)
        self.y2 = torch.empty((B, cap, C2), dtype=f32, device=dev)
        self.y3 = torch.empty((B, cap, C3), dtype=f32, device=dev)
        self.stat = torch.zeros((16 + rep * (2 * C2 + 2 * C3),), dtype=f64, device=dev)
        self.bn = torch.empty((4 * (C1 + C2 + C3),), dtype=f32, device=dev)
        self.gmom = torch.zeros((B * 12,), dtype=f64, device=dev)       # fcn_pn_group_compact: per-frustum moments + counter
        self.cnt = torch.empty((B, L), dtype=i32, device=dev)           # window hit counts of the fused grouping
        self.wenc = torch.empty((2 * (C2 * C1 + C3 * C2),), dtype=f32, device=dev)      # split-encoded conv2 / conv3 weights
        self.flags = flags if flags is not None else torch.zeros((1,), dtype=i32, device=dev)      # sticky FCN_FLAG_* bits
        # max-pool keys of conv3's epilogue (fcn_pn_ws.pkey, zero between launches): the pooling pass that re-reads y3 is replaced by
        # a pass over the (B, L, C3) keys.  Measured on MI355X: +6 % for the eval-mode forward (y3 is neither written nor
        # re-read).  For the TRAINING step round 3 measured it 0.5...1.5 % slower; behind rounds 4-5's changes it is 0.5 % (car), 1 %
        # (people), 2 % (SUN-RGBD) FASTER -- the PointNet forward phase ends 15 us earlier, the whole GPU suite passes with it -- and
        # 0.75 % slower on the refine configuration (L = 20 ... 3: the key pass is one more small launch on a chain of small
        # launches): keys by default from 65 536 window slots per scale up, FCN_POOL_KEYS=1 / 0 forces them on / off.
        mode = os.environ.get("FCN_POOL_KEYS")
        use_keys = mode == "1" or (mode not in ("0", "1") and (not need_grad or B * L * K >= 65536))
        self.pkey = torch.zeros((B, L, C3), dtype=torch.int64, device=dev) if use_keys else None
        self.amax = self.gmax = self.dy3 = self.dz2 = self.bstat = self.coef = self.partial = None
        self.nsplit = 0
        if need_grad:
            self.nsplit = ntile_max
            self.amax = torch.empty((B, L, C3), dtype=i32, device=dev)
            self.gmax = torch.empty((B, L, C3), dtype=f32, device=dev)
            # dy3 (B, cap, C3): written by the data-gradient GEMM of conv3, read back by its weight-gradient GEMM.  FCN_STORE_DY3=0
            # drops the buffer (a third of a scale's backward workspace: 0.9 GB over the four car scales at B = 32): the weight-
            # gradient GEMM then rebuilds dy3 while staging, bit-identically -- measured 0.7 % slower over the step
            if os.environ.get("FCN_STORE_DY3", "1") != "0":
                self.dy3 = torch.empty((B, cap, C3), dtype=f32, device=dev)
            self.dz2 = torch.empty((B, cap, C2), dtype=f32, device=dev)
            self.bstat = torch.zeros((rep * (2 * C3 + 2 * C2 + 4 * C1),), dtype=f64, device=dev)
            self.coef = torch.empty((5 * (C3 + C2),), dtype=f32, device=dev)
            self.partial = torch.empty((self.nsplit * (C3 * C2 + C2 * C1),), dtype=f32, device=dev)     # both weight gradients at once
        p = lambda t: None if t is None else t.data_ptr()
        self.c = PnWs(p(self.woff), p(self.ent), p(self.ewin), p(self.tiles), p(self.y2), p(self.y3), p(self.amax),
                      p(self.stat), p(self.bn), p(self.gmax), p(self.dy3), p(self.dz2), p(self.bstat),
                      p(self.coef), p(self.partial), self.nsplit, p(self.gmom), p(self.wenc), p(self.flags), p(self.pkey),
                      # `partial` above holds both weight gradients' split partials, so a one-stream backward can run conv2's data
                      # gradient and the two weight-gradient GEMMs as roles of ONE launch (pn_mid_kernel: 6 launches per scale instead
                      # of 8, bit-identical gradients) -- _mid_launch() below says where that pays
                      (1 if _mid_launch(B, L, K, C3) else (0 if os.environ.get("FCN_PN_TAIL", "1") == "0" else 2)) if need_grad else 0)

    @staticmethod
    def stored(t, precision_code):
        """fp32 view of one of the big intermediates (y2, y3, dy3, dz2) as the kernels of `precision_code` stored it: fp32, or --
        in the bf16 throughput mode (FCN_PREC_BF16 = 2) -- bf16 in the first half of the same buffer."""
        if precision_code != _precision.CODES["bf16"]:        # ("bf16ops" keeps fp32 storage)
            return t
        return t.view(-1).view(torch.bfloat16)[:t.numel()].view(t.shape).float()


class WorkspacePool:
    def __init__(self):
        self.free = {}
        self.side = {}
        self.side_wgrad = False     # weight-gradient GEMMs of the backward on a second stream (fcn_pn_backward2)
        self._flags = {}

    def flags(self, device):
        """Sticky FCN_FLAG_* bits of every workspace of this pool on `device` (one int32 tensor, zeroed at creation)."""
        key = str(device)
        if key not in self._flags:
            self._flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._flags[key]

    def side_stream(self, device):
        """Second HIP stream + 3 events (caller-owned, handed to fcn_pn_backward2) per device."""
        key = str(device)
        if key not in self.side:
            with torch.cuda.device(device):
                st = torch.cuda.Stream(device=device)
                st3 = torch.cuda.Stream(device=device)
                evs = [torch.cuda.Event(enable_timing=False) for _ in range(4)]
                for ev in evs:
                    ev.record()
                arr = (ctypes.c_void_p * 4)(*[ev.cuda_event for ev in evs])
            self.side[key] = (st, evs, arr, st3)
        return self.side[key]

    def bwd_events(self, device):
        key = "bwdev" + str(device)
        if key not in self.side:
            self.side[key] = (torch.cuda.Event(enable_timing=False), torch.cuda.Event(enable_timing=False))
        return self.side[key]

    def acquire(self, *key, device, need_grad):
        k = tuple(key) + (bool(need_grad), str(device))
        lst = self.free.setdefault(k, [])
        if lst:
            return lst.pop()
        ws = Workspace(*key, device=device, need_grad=need_grad, flags=self.flags(device))
        ws.pool_key = k
        return ws

    def release(self, ws):
        self.free.setdefault(ws.pool_key, []).append(ws)


def _params_struct(Ws, gammas, betas, rmeans, rvars, nbts):
    arr = lambda ts: (ctypes.c_void_p * 3)(*[None if t is None else t.data_ptr() for t in ts])
    return PnParams(arr(Ws), arr(gammas), arr(betas), arr(rmeans), arr(rvars), arr(nbts))


def _acquire(pool, cfgt, pc, ref, one_hot, bufs, plist, need_grad):
    """Workspace, descriptor and C parameter struct of one scale's forward (no launch)."""
    dist, K, training, eps, momentum = cfgt[:5]
    nlc = bool(cfgt[6]) if len(cfgt) > 6 else False
    W1, g1, b1, W2, g2, b2, W3, g3, b3 = plist
    B, _, N = pc.shape
    Lw = ref.shape[2]
    C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
    nvec = 0 if (one_hot is None or nlc) else one_hot.shape[1]
    dev = pc.device
    ws = pool.acquire(B, N, Lw, K, C1, C2, C3, device=dev, need_grad=need_grad)
    desc = PnDesc(B, N, Lw, K, C1, C2, C3, nvec, 1 if training else 0, eps, momentum, 1 if nlc else 0,
                  _precision.code(), 0)
    rmeans, rvars, nbts = bufs
    Wc = [W1.detach().reshape(C1, 3).contiguous(), W2.detach().reshape(C2, C1).contiguous(),
          W3.detach().reshape(C3, C2).contiguous()]
    gs = [g1.detach().contiguous(), g2.detach().contiguous(), g3.detach().contiguous()]
    bs = [b1.detach().contiguous(), b2.detach().contiguous(), b3.detach().contiguous()]
    params = _params_struct(Wc, gs, bs, rmeans, rvars, nbts)
    oh = None if (one_hot is None or nlc) else one_hot.detach().contiguous().float()
    return {"ws": ws, "desc": desc, "params": params, "Wc": Wc, "gs": gs, "bs": bs, "oh": oh, "nlc": nlc, "nvec": nvec,
            "ref": ref, "dist": float(dist), "dims": (B, Lw, C3), "dev": dev,
            "bufs": bufs}       # (the C struct holds raw pointers to the running statistics: the handle keeps their tensors alive)


def _run_forward(h, cnt, idx):
    """fcn_pn_forward of a prepared scale on the current stream -> the tuple _PointNetPooled keeps."""
    L = _native.lib()
    B, Lw, C3 = h["dims"]
    dev = h["dev"]
    feat = torch.empty((B, Lw, C3) if h["nlc"] else (B, C3 + h["nvec"], Lw), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(L.fcn_pn_forward(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), cnt.data_ptr(),
                                       None if h["oh"] is None else h["oh"].data_ptr(), ctypes.byref(h["ws"].c),
                                       feat.data_ptr(), _native.current_stream(dev)), "fcn_pn_forward")
    return feat, idx, cnt, h["ws"], h["desc"], (h["Wc"], h["gs"], h["bs"], cnt, idx, h["oh"])


def _forward_impl(pool, cfgt, pc, ref, one_hot, bufs, plist, need_grad):
    """API-form grouping (int64 idx) -> compaction -> fused forward of one scale; returns the workspace still held."""
    L = _native.lib()
    h = _acquire(pool, cfgt, pc, ref, one_hot, bufs, plist, need_grad)
    idx, cnt = query_depth_point(h["dist"], h["desc"].K, pc, ref)
    with torch.cuda.device(h["dev"]):
        _native.check(L.fcn_pn_compact(ctypes.byref(h["desc"]), pc.data_ptr(), ref.data_ptr(), idx.data_ptr(),
                                       cnt.data_ptr(), ctypes.byref(h["ws"].c), _native.current_stream(h["dev"])), "fcn_pn_compact")
    return _run_forward(h, cnt, idx)


def group_compact(handles, pc, phase=3):
    """The fused front of every prepared scale (fcn_pn_group_compact2): grouping + compaction + tile lists + input moments
    (phase 1: functions of the batch alone -- PointNetFeat.prefetch runs it for the NEXT batch beside the current step) and
    weight images + BN1 fold (phase 2: functions of this step's weights); phase 3 = both, two launches.  The int64 idx of the API
    form is never materialised.  Marks the descriptors `grouped` once the weight-dependent part has run."""
    L = _native.lib()
    n = len(handles)
    dev = pc.device
    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
    descs = arr([ctypes.addressof(h["desc"]) for h in handles])
    params = arr([ctypes.addressof(h["params"]) for h in handles])
    refs = arr([h["ref"].data_ptr() for h in handles])
    wss = arr([ctypes.addressof(h["ws"].c) for h in handles])
    cnts = arr([h["ws"].cnt.data_ptr() for h in handles])
    dz = (ctypes.c_float * n)(*[h["dist"] for h in handles])
    with torch.cuda.device(dev):
        _native.check(L.fcn_pn_group_compact2(n, descs, params, pc.data_ptr(), refs, dz, wss, cnts, int(phase),
                                              _native.current_stream(dev)), "fcn_pn_group_compact2")
    if phase != 1:
        for h in handles:
            h["desc"].grouped = 1


def _empty_idx(dev):
    return torch.empty((0,), dtype=torch.int64, device=dev)


class _PointNetPooled(torch.autograd.Function):
    """feat = pooled PointNet features.  Differentiable w.r.t. the 9 parameter tensors only (the reference
    never needs gradients w.r.t. the point cloud either: inputs do not require grad)."""

    @staticmethod
    def forward(ctx, pool, cfgt, pc, ref, one_hot, bufs, gdst, launched, W1, g1, b1, W2, g2, b2, W3, g3, b3):
        plist = (W1, g1, b1, W2, g2, b2, W3, g3, b3)
        ctx.gdst = gdst
        need_grad = bool(cfgt[5])     # decided by the caller: grad mode is always off inside Function.forward
        if launched is not None:      # kernels already enqueued by launch_pooled(); this call only creates the node
            feat, idx, cnt, ws, desc, keep = launched
        else:
            feat, idx, cnt, ws, desc, keep = _forward_impl(pool, cfgt, pc, ref, one_hot, bufs, plist, need_grad)
        ctx.pool = pool
        ctx.step = getattr(pool, "fwd_step", None)        # which forward of the step loop this node belongs to (PointNetFeat.forward)
        ctx.live = need_grad
        if need_grad:
            ctx.ws, ctx.desc, ctx.keep = ws, desc, keep
            ctx.shapes = (W1.shape, W2.shape, W3.shape)
        else:
            pool.release(ws)
        ctx.mark_non_differentiable(idx, cnt)
        ctx.set_materialize_grads(False)          # no zero-filled "gradients" for idx / cnt (two fill kernels per scale)
        return feat, idx, cnt

    @staticmethod
    def backward(ctx, dfeat, _didx, _dcnt):
        if dfeat is None:
            return (None,) * 17
        if not ctx.live:
            raise RuntimeError("fused PointNet forward ran without saved state (eval mode or no_grad)")
        L = _native.lib()
        ws, desc = ctx.ws, ctx.desc
        Wc, gs, bs, cnt, idx, oh = ctx.keep
        dev = dfeat.device
        C1, C2, C3 = desc.C1, desc.C2, desc.C3
        dfeat.record_stream(torch.cuda.current_stream(dev))
        dfeat = dfeat.contiguous().float()
        # gradient destinations: the parameter's flat-buffer view when the caller trains through FlatTrainState
        # (written in place, autograd gets None), else a fresh tensor handed back to autograd
        gd = ctx.gdst
        pick = lambda j, like: gd[j] if gd[j] is not None else torch.empty_like(like)
        dW = [pick(3 * i, Wc[i]) for i in range(3)]
        dg = [pick(3 * i + 1, gs[i]) for i in range(3)]
        db = [pick(3 * i + 2, bs[i]) for i in range(3)]
        params = _params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        three = False
        if ctx.pool.side_wgrad:
            side, _evs, evarr, side3 = ctx.pool.side_stream(dev)
            s2 = ctypes.c_void_p(side.cuda_stream)
            # three-way split (conv2's weight gradient on a stream of its own, fcn_pn_backward3): the widest scale's chain 335 -> 269 us,
            # but both narrow scales then wait for those branches and run alone at the tail -- step +4 % slower on MI355X (EXPERIMENTS.md
            # round 4).  pool.side_three turns it on (tests: bit-identical gradients)
            three = bool(getattr(ctx.pool, "side_three", False)) and hasattr(L, "fcn_pn_backward3")
        else:
            s2, evarr = None, None
        # bwd_stream (PointNetFeat.share_backward_stream): this scale's backward is enqueued on ANOTHER scale's stream, behind that
        # scale's chain -- one parallel branch less in the captured step's backward.  Eager: the host stream waits for this node's
        # stream (the gradient's producer) and this node's stream for the host's afterwards -- correct for any graph.  Under hipGraph
        # capture ROCm 7.2 cannot take an edge between two FORKED streams (the capture dumps core), so none is made: the host stream
        # must already be ordered behind the producer of `dfeat` -- PointNetFeat guarantees it (see share_backward_stream) -- and the
        # work is joined with the host scale's own (autograd joins every leaf stream into the caller's at the end of backward()).
        alt = getattr(ctx.pool, "bwd_stream", None)
        home = torch.cuda.current_stream(dev)
        capturing = False
        hostp = getattr(ctx.pool, "bwd_host_pool", None)
        if alt is not None and (hostp is None or getattr(hostp, "bwd_step", None) != ctx.step):
            alt = None                 # the host scale has not been differentiated for this forward (yet): own stream
        if alt is not None and alt != home:
            ev_in, ev_out = ctx.pool.bwd_events(dev)
            # (a real stream has a non-null handle; the host emulation of tests/ runs with inert stream objects and no device)
            capturing = bool(getattr(alt, "cuda_stream", 0)) and torch.cuda.is_current_stream_capturing()
            if capturing:
                cap = _native.capture_id(dev)
                with torch.cuda.stream(alt):
                    same = _native.capture_id(dev) == cap
                if not same:           # the host stream carries no work of THIS capture (its scale was differentiated in another
                    alt = None         # graph, or not at all): this scale stays on its own stream
            else:
                ev_in.record(home)
                alt.wait_event(ev_in)
            if alt is not None:
                dfeat.record_stream(alt)
        else:
            alt = None
        with torch.cuda.device(dev), torch.cuda.stream(alt if alt is not None else home):
            if three:
                _native.check(L.fcn_pn_backward3(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(),
                                                 ctypes.byref(ws.c), arr(dW), arr(dg), arr(db),
                                                 _native.current_stream(dev), s2, ctypes.c_void_p(side3.cuda_stream), evarr),
                              "fcn_pn_backward3")
            else:
                _native.check(L.fcn_pn_backward2(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(),
                                                 ctypes.byref(ws.c), arr(dW), arr(dg), arr(db),
                                                 _native.current_stream(dev), s2, evarr),
                              "fcn_pn_backward2")
            if alt is not None and not capturing:
                ev_out.record(alt)
        if alt is not None and not capturing:
            home.wait_event(ev_out)
        ctx.pool.bwd_step = ctx.step       # (this scale's backward of that forward is enqueued: it may host another scale's now)
        ctx.pool.release(ws)
        ctx.ws = None
        ctx.live = False
        s1, s2, s3 = ctx.shapes
        outs = [dW[0].view(s1), dg[0], db[0], dW[1].view(s2), dg[1], db[1], dW[2].view(s3), dg[2], db[2]]
        outs = [None if gd[j] is not None else t for j, t in enumerate(outs)]
        return (None, None, None, None, None, None, None, None) + tuple(outs)


def _cfg_tuple(dist, nsample, training, eps, momentum, params, nlc):
    need_grad = bool(training) and torch.is_grad_enabled() and any(t.requires_grad for t in params)
    return (float(dist), int(nsample), bool(training), float(eps), float(momentum), need_grad, bool(nlc))


def _check_device(pc):
    if not pc.is_cuda:
        raise RuntimeError("frustum_convnet_amd: the PointNet hot path runs on an MI355X only "
                           "(got a %s tensor); there is no CPU fallback" % pc.device)


def prepare_pooled(pool, dist, nsample, training, eps, momentum, pc, ref, one_hot, bufs, params, nlc=False):
    """Step 1 of the fused front: acquire this scale's workspace / descriptor.  Call group_compact() on the handles of all
    scales (one launch), then launch_prepared() per scale on its own stream."""
    _check_device(pc)
    cfgt = _cfg_tuple(dist, nsample, training, eps, momentum, params, nlc)
    h = _acquire(pool, cfgt, pc, ref, one_hot, bufs, params, cfgt[5])
    h["args"] = (cfgt, pc, ref, one_hot, bufs, params)
    return h


def launch_prepared(h):
    """fcn_pn_forward of a grouped scale on the current stream -> handle for attach_pooled()."""
    cfgt, pc, ref, one_hot, bufs, params = h["args"]
    with torch.no_grad():
        launched = _run_forward(h, h["ws"].cnt, _empty_idx(h["dev"]))
    return (cfgt, pc, ref, one_hot, bufs, params, launched)


def launch_pooled(pool, dist, nsample, training, eps, momentum, pc, ref, one_hot, bufs, params, nlc=False):
    """Enqueues the forward kernels of one scale on the current stream WITHOUT creating the autograd node and returns a
    handle for attach_pooled().  Splitting the two lets a caller launch the scales heaviest-first while creating their
    nodes lightest-first -- autograd runs backward nodes in reverse creation order, so the backward is heaviest-first too."""
    _check_device(pc)
    cfgt = _cfg_tuple(dist, nsample, training, eps, momentum, params, nlc)
    with torch.no_grad():
        launched = _forward_impl(pool, cfgt, pc, ref, one_hot, bufs, params, cfgt[5])
    return (cfgt, pc, ref, one_hot, bufs, params, launched)


def attach_pooled(pool, handle):
    """The differentiable output of a launch_pooled() handle (call it under the stream the kernels were launched on)."""
    cfgt, pc, ref, one_hot, bufs, params, launched = handle
    gdst = tuple(getattr(t, "_fcn_grad", None) for t in params)
    return _PointNetPooled.apply(pool, cfgt, pc, ref, one_hot, bufs, gdst, launched, *params)


def pointnet_pooled(pool, dist, nsample, training, eps, momentum, pc, ref, one_hot, bufs, params, nlc=False):
    """params = (W1,g1,b1,W2,g2,b2,W3,g3,b3); bufs = ([rm1,rm2,rm3],[rv1,rv2,rv3],[nbt1,nbt2,nbt3]).
    nlc=True returns position-major (B, L, C3) features without the one-hot rows (input of the fused ConvFeatNet)."""
    _check_device(pc)
    cfgt = _cfg_tuple(dist, nsample, training, eps, momentum, params, nlc)
    gdst = tuple(getattr(t, "_fcn_grad", None) for t in params)
    return _PointNetPooled.apply(pool, cfgt, pc, ref, one_hot, bufs, gdst, None, *params)


def _dense_view(ws, desc, cnt, dev):
    """(rows (B, L*K) int64: the entry row behind every slot -- slot k of window l maps to row woff[l] + (k < ne ? k : 0) --,
    a3 (B, cap, C3) = relu(bn3(y3)) of the entry rows, window mask (B, L)) of a finished forward."""
    B, L, K, C1, C2, C3 = desc.B, desc.L, desc.K, desc.C1, desc.C2, desc.C3
    off3 = 4 * (C1 + C2)
    s3, t3 = ws.bn[off3:off3 + C3], ws.bn[off3 + C3:off3 + 2 * C3]
    ne = cnt.clamp(min=1).long()
    k = torch.arange(K, device=dev).view(1, 1, K)
    rows = ws.woff[:, :L].long().unsqueeze(2) + torch.where(k < ne.unsqueeze(2), k, torch.zeros_like(k))
    a3 = torch.relu(ws.stored(ws.y3, desc.precision) * s3 + t3)
    return rows.view(B, L * K), a3, (cnt > 0)


def dense_from_entries(pool, dist, nsample, training, eps, momentum, pc, ref, bufs, params):
    """Reference-shaped (B, C3, L, K) masked activations of one scale (PointNetModule.forward's return,
    models/det_base.py:103), expanded from the per-entry conv3 output of the HIP forward.  Device-side indexing only; no
    autograd graph (dense_pointnet() is the differentiable form)."""
    if not pc.is_cuda:
        raise RuntimeError("frustum_convnet_amd: MI355X only; no CPU fallback")
    cfgt = (float(dist), int(nsample), bool(training), float(eps), float(momentum))
    feat, idx, cnt, ws, desc, _ = _forward_impl(pool, cfgt, pc, ref, None, bufs, params, False)
    B, L, K, C3 = desc.B, desc.L, desc.K, desc.C3
    rows, a3, live = _dense_view(ws, desc, cnt, pc.device)
    a = torch.gather(a3, 1, rows.unsqueeze(2).expand(-1, -1, C3)).view(B, L, K, C3) * live.view(B, L, 1, 1).float()
    out = a.permute(0, 3, 1, 2).contiguous()
    pool.release(ws)
    return out


class _PointNetDense(torch.autograd.Function):
    """PointNetModule.forward WITH its graph (models/det_base.py:62-103 returns a differentiable (B, C3, L, nsample) tensor): the
    forward is the entry-space HIP forward expanded to the dense slots; the backward sums the K slots of every window back onto
    its entry rows (the first hit collects its K - ne + 1 duplicates), applies the window and ReLU masks and hands the per-entry
    gradient to the HIP backward chain (fcn_pn_backward_dense: BatchNorm backward with multiplicities, both weight-gradient
    GEMMs, conv1 from its moments).  Differentiable w.r.t. the 9 parameter tensors."""

    @staticmethod
    def forward(ctx, pool, cfgt, pc, ref, bufs, *plist):
        feat, idx, cnt, ws, desc, keep = _forward_impl(pool, cfgt, pc, ref, None, bufs, plist, True)
        B, L, K, C3 = desc.B, desc.L, desc.K, desc.C3
        rows, a3, live = _dense_view(ws, desc, cnt, pc.device)
        a = torch.gather(a3, 1, rows.unsqueeze(2).expand(-1, -1, C3)).view(B, L, K, C3) * live.view(B, L, 1, 1).float()
        ctx.pool, ctx.ws, ctx.desc, ctx.keep = pool, ws, desc, keep
        ctx.rows, ctx.live = rows, live
        ctx.pos = a3 > 0
        ctx.shapes = tuple(t.shape for t in plist)
        return a.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, dout):
        L_ = _native.lib()
        ws, desc = ctx.ws, ctx.desc
        if ws is None:
            raise RuntimeError("the dense PointNet module's backward ran twice (its workspace is released after the first)")
        Wc, gs, bs = ctx.keep[0], ctx.keep[1], ctx.keep[2]
        B, L, K, C1, C2, C3 = desc.B, desc.L, desc.K, desc.C1, desc.C2, desc.C3
        cap, dev = L * K, dout.device
        # (B, C3, L, K) -> slots (B, L*K, C3), empty windows masked, summed onto the entry rows, ReLU mask
        g = (dout.float() * ctx.live.view(B, 1, L, 1).float()).permute(0, 2, 3, 1).reshape(B, L * K, C3)
        dz3 = torch.zeros((B, cap, C3), dtype=torch.float32, device=dev)
        dz3.scatter_add_(1, ctx.rows.unsqueeze(2).expand(-1, -1, C3), g)
        dz3 = torch.where(ctx.pos, dz3, torch.zeros_like(dz3)).contiguous()
        # the BatchNorm-backward sums of layer 3 (fcn_pn_ws.bstat, replica 0; the forward left the buffer zeroed)
        bn = ws.bn
        off3 = 4 * (C1 + C2)
        mean3, rstd3 = bn[off3 + 2 * C3:off3 + 3 * C3].double(), bn[off3 + 3 * C3:off3 + 4 * C3].double()
        y3 = ws.stored(ws.y3, desc.precision)
        d64 = dz3.double()
        ws.bstat.zero_()
        ws.bstat[:C3] = d64.sum(dim=(0, 1))
        # (rows past a frustum's live entries hold whatever the allocator left: dz3 is zero there, the product must be too)
        ws.bstat[C3:2 * C3] = torch.where(dz3 != 0, d64 * ((y3.double() - mean3) * rstd3), torch.zeros_like(d64)).sum(dim=(0, 1))
        dW = [torch.empty_like(w) for w in Wc]
        dg = [torch.empty_like(t) for t in gs]
        db = [torch.empty_like(t) for t in bs]
        params = _params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        with torch.cuda.device(dev):
            _native.check(L_.fcn_pn_backward_dense(ctypes.byref(desc), ctypes.byref(params), dz3.data_ptr(), ctypes.byref(ws.c),
                                                   arr(dW), arr(dg), arr(db), _native.current_stream(dev)), "fcn_pn_backward_dense")
        dz3.record_stream(torch.cuda.current_stream(dev))
        ctx.pool.release(ws)
        ctx.ws = None
        outs = []
        for i in range(3):
            outs += [dW[i].view(ctx.shapes[3 * i]), dg[i], db[i]]
        return (None, None, None, None, None) + tuple(outs)


def dense_pointnet(pool, dist, nsample, training, eps, momentum, pc, ref, bufs, params):
    """Differentiable PointNetModule.forward: (B, C3, L, K) masked activations carrying a graph to the 9 parameter tensors."""
    _check_device(pc)
    if os.environ.get("FCN_STORE_DY3", "1") == "0":
        raise RuntimeError("the differentiable dense module API needs the dy3 buffer (unset FCN_STORE_DY3=0)")
    cfgt = _cfg_tuple(dist, nsample, training, eps, momentum, params, False)
    return _PointNetDense.apply(pool, cfgt, pc, ref, bufs, *params)
