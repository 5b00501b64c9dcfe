"""Autograd binding of the fused ConvFeatNet + heads (C-ABI fcn_convnet_forward / fcn_convnet_backward,
csrc/fcn_net.hip).  Input: the four (models/det_base.py) or five (models/det_base_sunrgbd.py) position-major pooled
feature maps of the PointNet scales; output: row-major logits (B*L2, ld) (cols 0..1 cls_out, 2.. reg_out; ld = 64 for
KITTI's 41 columns, 128 for SUN-RGBD's 69) that feed the fused loss tail directly."""
import ctypes
import os

import torch

from . import _native
from ._native import CnDesc, CnParams, CnWs, CN_MAXLEV, CN_MAXLAYER
from . import precision as _precision
from .common import bn_momentum

def layer_names(nlev):
    """BN layers of ConvFeatNet in the C-ABI's order (include/fcn_hip.h, fcn_cn_params): block1_conv1, block{j}_conv1 /
    _conv2 / _merge for j = 2..nlev, then block{j}_deconv for j = 2..nlev (13 names for 4 levels, 17 for 5)."""
    names = ["block1_conv1"]
    for j in range(2, nlev + 1):
        names += ["block%d_conv1" % j, "block%d_conv2" % j, "block%d_merge" % j]
    names += ["block%d_deconv" % j for j in range(2, nlev + 1)]
    return tuple(names)


LAYERS = layer_names(4)


def num_levels(conv_net):
    return 5 if hasattr(conv_net, "block5_conv1") else 4


class CnWorkspace:
    def __init__(self, desc, device, need_grad, flags=None):
        L = _native.lib()
        sizes = (ctypes.c_int64 * 6)()
        _native.check(L.fcn_convnet_sizes(ctypes.byref(desc), ctypes.byref(sizes)), "fcn_convnet_sizes")
        ny, nwp, nbn, nst, ncoef, npart = [int(v) for v in sizes]
        f32, f64 = torch.float32, torch.float64
        self.y = torch.empty(ny, dtype=f32, device=device)
        self.wp = torch.empty(nwp, dtype=f32, device=device)
        self.bn = torch.empty(nbn, dtype=f32, device=device)
        self.stat = torch.zeros(nst, dtype=f64, device=device)
        self.partial = torch.empty(npart, dtype=f32, device=device)
        self.oh64 = torch.zeros(desc.B * 64, dtype=f32, device=device)
        self.dz = self.bstat = self.coef = None
        if need_grad:
            self.dz = torch.empty(ny, dtype=f32, device=device)
            self.bstat = torch.zeros(nst, dtype=f64, device=device)
            self.coef = torch.empty(ncoef, dtype=f32, device=device)
        self.flags = flags if flags is not None else torch.zeros(1, dtype=torch.int32, device=device)
        p = lambda t: None if t is None else t.data_ptr()
        self.c = CnWs(p(self.y), p(self.dz), p(self.wp), p(self.bn), p(self.stat), p(self.bstat), p(self.coef),
                      p(self.partial), p(self.oh64), p(self.flags))


class CnPool:
    def __init__(self):
        self.free = {}
        self.side = {}
        self._flags = {}

    def flags(self, device):
        """Sticky FCN_FLAG_* bits of every workspace of this pool on `device` (one int32 tensor, zeroed at creation)."""
        key = str(device)
        if key not in self._flags:
            self._flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._flags[key]

    def pack_stream(self, device):
        """Side stream + event for the early weight packing (one per device)."""
        key = str(device)
        if key not in self.side:
            with torch.cuda.device(device):
                self.side[key] = (torch.cuda.Stream(device=device), torch.cuda.Event(enable_timing=False))
        return self.side[key]

    def acquire(self, key, desc, device, need_grad):
        k = key + (bool(need_grad), str(device))
        lst = self.free.setdefault(k, [])
        if lst:
            return lst.pop()
        ws = CnWorkspace(desc, device, need_grad, flags=self.flags(device))
        ws.pool_key = k
        return ws

    def release(self, ws):
        self.free.setdefault(ws.pool_key, []).append(ws)


def _arr(ts, n=CN_MAXLAYER):
    vals = [None if t is None else t.data_ptr() for t in ts] + [None] * (n - len(ts))
    return (ctypes.c_void_p * n)(*vals)


def _prepare(pool, cfgt, bufs, one_hot, B, Ls, dev, pt):
    """Detached parameter views, descriptor, workspace and the C parameter struct of one forward."""
    training, eps, momentum, need_grad = cfgt
    nlev = len(Ls)
    nb = 4 * nlev - 3                          # BN layers: 13 (4 levels) / 17 (5 levels)
    Ws = [w.detach().contiguous() for w in pt[0:nb]]
    gs = [g.detach().contiguous() for g in pt[nb:2 * nb]]
    bs = [b.detach().contiguous() for b in pt[2 * nb:3 * nb]]
    cls_w, reg_w, cls_b, reg_b = [t.detach() for t in pt[3 * nb:3 * nb + 4]]
    # heads as one (2 + reg_out, 256 * (nlev - 1)) matrix: adjacent in memory under FlatTrainState (no copy), else concatenated
    Wh = _adjacent(cls_w, reg_w)
    if Wh is None:
        Wh = torch.cat([cls_w, reg_w], 0).contiguous()
    bh = _adjacent(cls_b, reg_b)
    if bh is None:
        bh = torch.cat([cls_b, reg_b], 0).contiguous()
    nvec = 0 if one_hot is None else one_hot.shape[1]
    oh = None if one_hot is None else one_hot.detach().contiguous().float()
    desc = CnDesc(B, (ctypes.c_int32 * CN_MAXLEV)(*Ls), nvec, reg_w.shape[0], 1 if training else 0, eps, momentum, 0,
                  _precision.code(), nlev, Ws[0].shape[0])
    ws = pool.acquire((B,) + tuple(Ls) + (nvec, reg_w.shape[0]), desc, dev, need_grad)
    rmeans, rvars, nbts = bufs
    params = CnParams(_arr(Ws + [Wh]), _arr(gs), _arr(bs), _arr(rmeans), _arr(rvars), _arr(nbts), bh.data_ptr())
    return {"Ws": Ws, "gs": gs, "bs": bs, "Wh": Wh, "bh": bh, "oh": oh, "desc": desc, "ws": ws, "params": params,
            "event": None}


class _ConvNetFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pool, cfgt, bufs, one_hot, gdst, pre, feat_events, nlev, *rest):
        # rest: nlev pooled feature maps, then pt = nb conv weights, nb gammas, nb betas (nb = 4 * nlev - 3), cls_w, reg_w,
        # cls_b, reg_b
        training, eps, momentum, need_grad = cfgt
        ctx.gdst = gdst
        ctx.nlev = nlev
        nb = 4 * nlev - 3
        L = _native.lib()
        feats = [f.detach().contiguous() for f in rest[:nlev]]
        pt = rest[nlev:]
        B = feats[0].shape[0]
        Ls = [f.shape[1] for f in feats]
        dev = feats[0].device
        if pre is None:
            pre = _prepare(pool, cfgt, bufs, one_hot, B, Ls, dev, pt)
        elif pre["event"] is not None:
            torch.cuda.current_stream(dev).wait_event(pre["event"])      # the packing ran on the side stream
        Ws, gs, bs, Wh, bh, oh, desc, ws, params = (pre[k] for k in ("Ws", "gs", "bs", "Wh", "bh", "oh", "desc", "ws",
                                                                    "params"))
        assert list(desc.L)[:nlev] == Ls and desc.B == B
        ld = L.fcn_convnet_logits_ld(ctypes.byref(desc))
        if ld <= 0:
            raise _native.NativeError("fcn_convnet_logits_ld: unsupported head width %d" % desc.reg_out)
        logits = torch.empty((B * Ls[1], ld), dtype=torch.float32, device=dev)
        fp = _arr(feats, CN_MAXLEV)
        evarr = None
        if feat_events is not None:      # per-map completion events: the C side waits for each right before its first use
            evarr = (ctypes.c_void_p * CN_MAXLEV)(*([None if e is None else e.cuda_event for e in feat_events] +
                                                    [None] * (CN_MAXLEV - len(feat_events))))
            cur = torch.cuda.current_stream(dev)
            for ft in feats:
                ft.record_stream(cur)    # produced on the scales' streams, consumed here
        with torch.cuda.device(dev):
            _native.check(L.fcn_convnet_forward2(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(ws.c), fp,
                                                 None if oh is None else oh.data_ptr(), logits.data_ptr(),
                                                 _native.current_stream(dev), evarr), "fcn_convnet_forward2")
        ctx.pool, ctx.live = pool, need_grad
        if need_grad:
            ctx.ws, ctx.desc, ctx.keep = ws, desc, (feats, oh, Ws, Wh, gs, bs, bh)
            ctx.shapes = [w.shape for w in pt[0:nb]]
        else:
            pool.release(ws)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if not ctx.live:
            raise RuntimeError("fused ConvFeatNet forward ran without saved state (eval mode or no_grad)")
        L = _native.lib()
        ws, desc = ctx.ws, ctx.desc
        feats, oh, Ws, Wh, gs, bs, bh = ctx.keep
        dev = dlogits.device
        dlogits = dlogits.contiguous().float()
        nlev = ctx.nlev
        nb = 4 * nlev - 3
        dfeats = [torch.empty_like(f) for f in feats]
        # gradient destinations: flat-buffer views under FlatTrainState (written in place, autograd gets None)
        gd = ctx.gdst
        pick = lambda j, like: gd[j] if gd[j] is not None else torch.empty_like(like)
        h0 = 3 * nb                               # cls_w, reg_w, cls_b, reg_b follow the 3 * nb conv / BN parameters
        dWh = None if gd[h0] is None or gd[h0 + 1] is None else _adjacent(gd[h0], gd[h0 + 1])
        dbh = None if gd[h0 + 2] is None or gd[h0 + 3] is None else _adjacent(gd[h0 + 2], gd[h0 + 3])
        heads_direct = dWh is not None and dbh is not None
        if not heads_direct:
            dWh, dbh = torch.empty_like(Wh), torch.empty_like(bh)
        dW = [pick(i, Ws[i]) for i in range(nb)] + [dWh]
        dg = [pick(nb + i, gs[i]) for i in range(nb)]
        db = [pick(2 * nb + i, bs[i]) for i in range(nb)]
        params = CnParams(_arr(Ws + [Wh]), _arr(gs), _arr(bs), _arr([]), _arr([]), _arr([]), bh.data_ptr())
        fp = _arr(feats, CN_MAXLEV)
        dfp = _arr(dfeats, CN_MAXLEV)
        # (the C-ABI can continue the chain on a second stream once the widest map's gradient is final -- stream2 / events of
        # fcn_convnet_backward; measured slower over the step on MI355X in two rounds, EXPERIMENTS.md, so this layer passes none)
        with torch.cuda.device(dev):
            _native.check(L.fcn_convnet_backward(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(ws.c), fp,
                                                 None if oh is None else oh.data_ptr(), dlogits.data_ptr(), dfp,
                                                 _arr(dW), _arr(dg), _arr(db), dbh.data_ptr(),
                                                 _native.current_stream(dev), None, None),
                          "fcn_convnet_backward")
        ctx.pool.release(ws)
        ctx.ws, ctx.live = None, False
        ncls = 2
        outs = list(dW[:nb]) + dg + db
        outs = [None if gd[j] is not None else t for j, t in enumerate(outs)]
        if heads_direct:
            hz = [None] * 4
        else:
            hz = [dW[nb][:ncls], dW[nb][ncls:], dbh[:ncls], dbh[ncls:]]
            for j in range(4):
                if gd[h0 + j] is not None:           # flat views that are not adjacent: copy in, hand autograd nothing
                    gd[h0 + j].copy_(hz[j])
                    hz[j] = None
        return (None,) * 8 + tuple(dfeats) + tuple(outs) + tuple(hz)


def _adjacent(a, b):
    """One tensor over a and b when b starts where a ends in memory (rows of the same width), else None."""
    if not (a.is_contiguous() and b.is_contiguous()) or a.shape[1:] != b.shape[1:]:
        return None
    if b.data_ptr() != a.data_ptr() + a.numel() * a.element_size():
        return None
    try:
        return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
    except RuntimeError:
        return None


def _gather(conv_net, cls_out, reg_out):
    seqs = [getattr(conv_net, n) for n in layer_names(num_levels(conv_net))]
    Ws = [s[0].weight for s in seqs]
    gs = [s[1].weight for s in seqs]
    bs = [s[1].bias for s in seqs]
    bufs = ([s[1].running_mean for s in seqs], [s[1].running_var for s in seqs],
            [s[1].num_batches_tracked for s in seqs])
    bn0 = seqs[0][1]
    pt = Ws + gs + bs + [cls_out.weight, reg_out.weight, cls_out.bias, reg_out.bias]
    return pt, bufs, bn0


def convnet_prepack(pool, conv_net, cls_out, reg_out, B, Ls, one_hot, device):
    """Starts the weight re-packing of the coming convnet_fused() call on the pool's side stream (forked from the current
    stream) and returns the handle to pass as `pre`: 25 us that overlap the PointNet scales instead of heading the FCN."""
    pt, bufs, bn0 = _gather(conv_net, cls_out, reg_out)
    training = conv_net.training
    need_grad = bool(training) and torch.is_grad_enabled() and any(t.requires_grad for t in pt)
    cfgt = (bool(training), float(bn0.eps), bn_momentum(bn0), need_grad)
    cur = torch.cuda.current_stream(device)
    side, ev = pool.pack_stream(device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        # a step loop may hang work in FRONT of the packing, on its stream (bench.py FCN_ADAM_LATE: the optimiser step of the
        # [ConvFeatNet + heads] bucket, whose result nothing reads before this packing)
        cb = getattr(pool, "before_pack", None)
        if cb is not None:
            cb()
        pre = _prepare(pool, cfgt, bufs, one_hot, B, list(Ls), device, pt)
        with torch.cuda.device(device):
            _native.check(_native.lib().fcn_convnet_pack(ctypes.byref(pre["desc"]), ctypes.byref(pre["params"]),
                                                         ctypes.byref(pre["ws"].c),
                                                         None if pre["oh"] is None else pre["oh"].data_ptr(),
                                                         _native.current_stream(device)), "fcn_convnet_pack")
        ev.record(side)
    pre["desc"].prepacked = 1
    pre["event"] = ev
    pre["cfgt"] = cfgt
    return pre


def convnet_fused(pool, conv_net, cls_out, reg_out, feats, one_hot, pre=None, feat_events=None):
    """feats: 4 or 5 x (B, L_s, C_s) position-major pooled features.  Returns logits (B*L2, 64 | 128).
    feat_events: per-map torch.cuda.Event recorded when the map is complete on its producer's stream (None: the maps
    are already ordered before the current stream)."""
    if not feats[0].is_cuda:
        raise RuntimeError("frustum_convnet_amd: fused ConvFeatNet runs on the GPU only")
    pt, bufs, bn0 = _gather(conv_net, cls_out, reg_out)
    training = conv_net.training
    need_grad = bool(training) and torch.is_grad_enabled() and (
        any(t.requires_grad for t in pt) or any(f.requires_grad for f in feats))
    cfgt = (bool(training), float(bn0.eps), bn_momentum(bn0), need_grad)
    if pre is not None and pre["cfgt"] != cfgt:        # e.g. only the features require grad: pack inline instead
        pool.release(pre["ws"])
        torch.cuda.current_stream(feats[0].device).wait_event(pre["event"])
        pre = None
    gdst = tuple(getattr(t, "_fcn_grad", None) for t in pt)
    if len(feats) != num_levels(conv_net):
        raise ValueError("ConvFeatNet with %d levels got %d feature maps" % (num_levels(conv_net), len(feats)))
    return _ConvNetFused.apply(pool, cfgt, bufs, one_hot, gdst, pre, feat_events, len(feats), *feats, *pt)
