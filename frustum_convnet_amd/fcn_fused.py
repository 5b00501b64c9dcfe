"""Autograd binding of the fused ConvFeatNet + heads (C-ABI fcn_convnet_forward / fcn_convnet_backward,
csrc/fcn_net.hip).  Input: the four position-major pooled feature maps of the PointNet scales; output: row-major
logits (B*L2, 64) (cols 0..1 cls_out, 2..40 reg_out) that feed the fused loss tail directly."""
import ctypes
import os

import torch

from . import _native
from ._native import CnDesc, CnParams, CnWs
from . import precision as _precision
from .common import bn_momentum

LAYERS = ("block1_conv1", "block2_conv1", "block2_conv2", "block2_merge", "block3_conv1", "block3_conv2",
          "block3_merge", "block4_conv1", "block4_conv2", "block4_merge", "block2_deconv", "block3_deconv",
          "block4_deconv")


class CnWorkspace:
    def __init__(self, desc, device, need_grad):
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
        p = lambda t: None if t is None else t.data_ptr()
        self.c = CnWs(p(self.y), p(self.dz), p(self.wp), p(self.bn), p(self.stat), p(self.bstat), p(self.coef),
                      p(self.partial), p(self.oh64))


# gradient tensor (data_ptr) -> event its consumer must wait for: set by _ConvNetFused.backward for the feature-map
# gradients that become final on the continuation stream, consumed by pointnet_fused._PointNetPooled.backward
PENDING_GRADS = {}


def wait_pending_grad(t):
    ev = PENDING_GRADS.pop(t.data_ptr(), None)
    if ev is not None:
        torch.cuda.current_stream(t.device).wait_event(ev)


class CnPool:
    def __init__(self):
        self.free = {}
        self.side = {}
        self.cont = {}
        self.last_done = None

    def cont_stream(self, device):
        """Continuation stream of the backward + its 4 events (caller-owned, handed to fcn_convnet_backward)."""
        key = str(device)
        if key not in self.cont:
            with torch.cuda.device(device):
                st = torch.cuda.Stream(device=device)
                evs = [torch.cuda.Event(enable_timing=False) for _ in range(4)]
                for ev in evs:
                    ev.record()                     # materialise the hipEvent_t handles
                arr = (ctypes.c_void_p * 4)(*[ev.cuda_event for ev in evs])
            self.cont[key] = (st, evs, arr)
        return self.cont[key]

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
        ws = CnWorkspace(desc, device, need_grad)
        ws.pool_key = k
        return ws

    def release(self, ws):
        self.free.setdefault(ws.pool_key, []).append(ws)


def _arr(ts, n=14):
    vals = [None if t is None else t.data_ptr() for t in ts] + [None] * (n - len(ts))
    return (ctypes.c_void_p * n)(*vals)


def _prepare(pool, cfgt, bufs, one_hot, B, Ls, dev, pt):
    """Detached parameter views, descriptor, workspace and the C parameter struct of one forward."""
    training, eps, momentum, need_grad = cfgt
    Ws = [w.detach().contiguous() for w in pt[0:13]]
    gs = [g.detach().contiguous() for g in pt[13:26]]
    bs = [b.detach().contiguous() for b in pt[26:39]]
    cls_w, reg_w, cls_b, reg_b = [t.detach() for t in pt[39:43]]
    # heads as one (2 + reg_out, 768) matrix: adjacent in memory under FlatTrainState (no copy), else concatenated
    Wh = _adjacent(cls_w, reg_w)
    if Wh is None:
        Wh = torch.cat([cls_w, reg_w], 0).contiguous()
    bh = _adjacent(cls_b, reg_b)
    if bh is None:
        bh = torch.cat([cls_b, reg_b], 0).contiguous()
    nvec = 0 if one_hot is None else one_hot.shape[1]
    oh = None if one_hot is None else one_hot.detach().contiguous().float()
    desc = CnDesc(B, (ctypes.c_int32 * 4)(*Ls), nvec, reg_w.shape[0], 1 if training else 0, eps, momentum, 0,
                  _precision.code())
    ws = pool.acquire((B,) + tuple(Ls) + (nvec, reg_w.shape[0]), desc, dev, need_grad)
    rmeans, rvars, nbts = bufs
    params = CnParams(_arr(Ws + [Wh]), _arr(gs), _arr(bs), _arr(rmeans), _arr(rvars), _arr(nbts), bh.data_ptr())
    return {"Ws": Ws, "gs": gs, "bs": bs, "Wh": Wh, "bh": bh, "oh": oh, "desc": desc, "ws": ws, "params": params,
            "event": None}


class _ConvNetFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pool, cfgt, bufs, one_hot, gdst, pre, feat_events, f1, f2, f3, f4, *pt):
        # pt: 13 conv weights, 13 gammas, 13 betas, cls_w, reg_w, cls_b, reg_b
        training, eps, momentum, need_grad = cfgt
        ctx.gdst = gdst
        L = _native.lib()
        feats = [f.detach().contiguous() for f in (f1, f2, f3, f4)]
        B = feats[0].shape[0]
        Ls = [f.shape[1] for f in feats]
        dev = feats[0].device
        if pre is None:
            pre = _prepare(pool, cfgt, bufs, one_hot, B, Ls, dev, pt)
        elif pre["event"] is not None:
            torch.cuda.current_stream(dev).wait_event(pre["event"])      # the packing ran on the side stream
        Ws, gs, bs, Wh, bh, oh, desc, ws, params = (pre[k] for k in ("Ws", "gs", "bs", "Wh", "bh", "oh", "desc", "ws",
                                                                    "params"))
        assert list(desc.L) == Ls and desc.B == B
        logits = torch.empty((B * Ls[1], 64), dtype=torch.float32, device=dev)
        fp = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
        evarr = None
        if feat_events is not None:      # per-map completion events: the C side waits for each right before its first use
            evarr = (ctypes.c_void_p * 4)(*[None if e is None else e.cuda_event for e in feat_events])
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
            ctx.shapes = [w.shape for w in pt[0:13]]
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
        dfeats = [torch.empty_like(f) for f in feats]
        # gradient destinations: flat-buffer views under FlatTrainState (written in place, autograd gets None)
        gd = ctx.gdst
        pick = lambda j, like: gd[j] if gd[j] is not None else torch.empty_like(like)
        dWh = None if gd[39] is None or gd[40] is None else _adjacent(gd[39], gd[40])
        dbh = None if gd[41] is None or gd[42] is None else _adjacent(gd[41], gd[42])
        heads_direct = dWh is not None and dbh is not None
        if not heads_direct:
            dWh, dbh = torch.empty_like(Wh), torch.empty_like(bh)
        dW = [pick(i, Ws[i]) for i in range(13)] + [dWh]
        dg = [pick(13 + i, gs[i]) for i in range(13)]
        db = [pick(26 + i, bs[i]) for i in range(13)]
        params = CnParams(_arr(Ws + [Wh]), _arr(gs), _arr(bs), _arr([]), _arr([]), _arr([]), bh.data_ptr())
        fp = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
        dfp = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in dfeats])
        # continuation stream: after the third launch dfeats[3] is final and the rest of the chain moves to a second
        # stream, so the scale-4 PointNet backward (the long pole, next on THIS stream) overlaps it.  The gradients of the
        # other maps become final on that stream: their consumers find the event to wait for in PENDING_GRADS.
        use_cont = bool(int(os.environ.get("FCN_TOPO", "0")) & 4)    # see det_base.PointNetFeat: measured slower, off
        cont, evs, evarr = ctx.pool.cont_stream(dev)
        if use_cont:
            for t in dfeats[:3]:
                t.record_stream(cont)
        with torch.cuda.device(dev):
            _native.check(L.fcn_convnet_backward(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(ws.c), fp,
                                                 None if oh is None else oh.data_ptr(), dlogits.data_ptr(), dfp,
                                                 _arr(dW), _arr(dg), _arr(db), dbh.data_ptr(),
                                                 _native.current_stream(dev),
                                                 ctypes.c_void_p(cont.cuda_stream) if use_cont else None,
                                                 evarr if use_cont else None),
                          "fcn_convnet_backward")
        if not use_cont:
            evs = [None] * 4
        else:
            PENDING_GRADS[dfeats[2].data_ptr()] = evs[1]
            PENDING_GRADS[dfeats[1].data_ptr()] = evs[2]
            PENDING_GRADS[dfeats[0].data_ptr()] = evs[3]
        # the parameter gradients are final at evs[3] as well: whoever consumes dfeats[0] joins the continuation stream,
        # and this stream joins it here when nobody will (no PointNet consumer, e.g. features without grad)
        ctx.pool.last_done = evs[3]
        ctx.pool.release(ws)
        ctx.ws, ctx.live = None, False
        ncls = 2
        outs = list(dW[:13]) + dg + db
        outs = [None if gd[j] is not None else t for j, t in enumerate(outs)]
        if heads_direct:
            hz = [None] * 4
        else:
            hz = [dW[13][:ncls], dW[13][ncls:], dbh[:ncls], dbh[ncls:]]
            for j in range(4):
                if gd[39 + j] is not None:           # flat views that are not adjacent: copy in, hand autograd nothing
                    gd[39 + j].copy_(hz[j])
                    hz[j] = None
        if evs[3] is not None and (any(o is not None for o in outs) or any(h is not None for h in hz)):
            # ordinary autograd gradients are consumed on THIS stream as soon as we return: they are final on the
            # continuation stream only (FlatTrainState's in-place gradients need no such wait)
            torch.cuda.current_stream(dev).wait_event(evs[3])
        return (None, None, None, None, None, None, None, dfeats[0], dfeats[1], dfeats[2], dfeats[3]) + tuple(outs) + \
            tuple(hz)


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
    seqs = [getattr(conv_net, n) for n in LAYERS]
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
    """feats: 4 x (B, L_s, C_s) position-major pooled features.  Returns logits (B*L2, 64).
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
    return _ConvNetFused.apply(pool, cfgt, bufs, one_hot, gdst, pre, feat_events, feats[0], feats[1], feats[2], feats[3],
                               *pt)
