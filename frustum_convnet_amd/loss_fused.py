"""Autograd binding of the fused train-loss tail (C-ABI fcn_det_loss_tail, csrc/loss_tail.hip).

Replaces the ~150 elementwise torch ops of models/det_base.py:373-476 by one launch that returns the 8 loss
scalars, 3 accuracies and d(total_loss)/d(logits).  Only `total_loss` is differentiable (the reference never
back-propagates the other entries either: train/train_net_det.py:124-127)."""
import torch

from . import _native


_UNIT = {}


def unit_grad(device):
    """A cached 0-dim tensor holding 1.0: pass it as the gradient seed (loss.backward(gradient=unit_grad(dev)), or use
    PointNetDet.backward) and the step contains neither autograd's ones-fill kernel nor the `dlogits * 1` multiply -- the fused
    tail recognises the seed by identity and hands its precomputed d total / d logits through untouched."""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT[key]


def _is_unit(g):
    u = _UNIT.get(str(g.device))
    return u is not None and g.data_ptr() == u.data_ptr()


def _check_labels(cls_label, size_class, ns):
    """The kernels read both label tensors as int64 through raw pointers: a wrong dtype would be misread silently.
    (size_class values must lie in [0, ns): the kernel clamps them, the reference would raise an index error.)"""
    if cls_label.dtype != torch.int64 or size_class.dtype != torch.int64:
        raise TypeError("cls_label / size_class must be int64 (got %s / %s)" % (cls_label.dtype, size_class.dtype))

LOSS_NAMES = ("total_loss", "cls_loss", "center_loss", "head_cls_loss", "head_res_loss", "size_cls_loss",
              "size_res_loss", "corners_loss")


class _LossTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_raw, reg_raw, cls_label, ref2, center, heading, size, size_class, mean_size, nb, ns, w):
        L = _native.lib()
        _check_labels(cls_label, size_class, ns)
        B, _, L2 = cls_raw.shape
        cls_c, reg_c = cls_raw.detach().contiguous(), reg_raw.detach().contiguous()
        need = cls_raw.requires_grad or reg_raw.requires_grad
        out = torch.empty(16, dtype=torch.float32, device=cls_raw.device)
        dcls = torch.empty_like(cls_c) if need else None
        dreg = torch.empty_like(reg_c) if need else None
        args = [cls_label.contiguous(), ref2.contiguous().float(), center.contiguous().float(),
                heading.contiguous().float(), size.contiguous().float(), size_class.contiguous(),
                mean_size.contiguous().float()]
        with torch.cuda.device(cls_raw.device):
            rc = L.fcn_det_loss_tail(cls_c.data_ptr(), reg_c.data_ptr(), *[t.data_ptr() for t in args],
                                     B, L2, int(nb), int(ns), float(w[0]), float(w[1]), float(w[2]), float(w[3]),
                                     out.data_ptr(), None if dcls is None else dcls.data_ptr(),
                                     None if dreg is None else dreg.data_ptr(), _native.current_stream(cls_raw.device))
        _native.check(rc, "fcn_det_loss_tail")
        ctx.save_for_backward(*(t for t in (dcls, dreg) if t is not None))
        ctx.need = need
        total = out[0].clone()
        rest = out.detach()
        ctx.mark_non_differentiable(rest)
        return total, rest

    @staticmethod
    def backward(ctx, gtotal, _grest):
        if not ctx.need:
            return (None,) * 12
        dcls, dreg = ctx.saved_tensors
        return (dcls * gtotal, dreg * gtotal) + (None,) * 10


def det_loss_tail(cls_raw, reg_raw, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size, size_class,
                  mean_size, num_bins, num_sizes, weights):
    """-> (losses dict with the reference's 8 keys, accuracies (cls, head, size), nfg)."""
    if not cls_raw.is_cuda:
        raise RuntimeError("frustum_convnet_amd: fused loss tail runs on the GPU only")
    total, rest = _LossTail.apply(cls_raw, reg_raw, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size,
                                  size_class, mean_size, num_bins, num_sizes, weights)
    losses = {"total_loss": total}
    for i, k in enumerate(LOSS_NAMES[1:], start=1):
        losses[k] = rest[i]
    return losses, (rest[8], rest[9], rest[10]), rest[11]


class _LossTailRows(torch.autograd.Function):
    """Same tail on the row-major (B*L2, 64) logits of the fused ConvFeatNet."""

    @staticmethod
    def forward(ctx, logits, cls_label, ref2, center, heading, size, size_class, mean_size, B, L2, nb, ns, w, scratch):
        L = _native.lib()
        _check_labels(cls_label, size_class, ns)
        lg = logits.detach().contiguous()
        need = logits.requires_grad
        out = torch.empty(16, dtype=torch.float32, device=lg.device)
        total = torch.empty((), dtype=torch.float32, device=lg.device)      # own storage: no clone of out[0]
        dlog = torch.empty_like(lg) if need else None
        args = [cls_label.contiguous(), ref2.contiguous().float(), center.contiguous().float(),
                heading.contiguous().float(), size.contiguous().float(), size_class.contiguous(),
                mean_size.contiguous().float()]
        with torch.cuda.device(lg.device):
            rc = L.fcn_det_loss_tail_rows2(lg.data_ptr(), *[t.data_ptr() for t in args], int(B), int(L2), int(nb),
                                           int(ns), float(w[0]), float(w[1]), float(w[2]), float(w[3]),
                                           out.data_ptr(), None if dlog is None else dlog.data_ptr(),
                                           None if scratch is None else scratch.data_ptr(), total.data_ptr(),
                                           _native.current_stream(lg.device))
        _native.check(rc, "fcn_det_loss_tail_rows2")
        ctx.need = need
        if need:
            ctx.save_for_backward(dlog)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)          # no zero-filled gradient for `out`
        return total, out

    @staticmethod
    def backward(ctx, gtotal, _grest):
        if not ctx.need or gtotal is None:
            return (None,) * 14
        (dlog,) = ctx.saved_tensors
        return (dlog if _is_unit(gtotal) else dlog * gtotal,) + (None,) * 13


def loss_scratch(B, L2, device):
    """Persistent scratch of the row-major loss tail (zeroed once; the kernel leaves it ready for the next launch).  One
    per module / stream: two launches sharing it must not overlap."""
    n = int(_native.lib().fcn_det_loss_tail_scratch_floats(int(B), int(L2)))
    return torch.zeros(max(n, 32), dtype=torch.float32, device=device)


def det_loss_tail_rows(logits, B, L2, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size, size_class,
                       mean_size, num_bins, num_sizes, weights, scratch=None):
    """-> (losses, (cls_acc, head_acc, size_acc), nfg) -- all device scalars."""
    total, rest = _LossTailRows.apply(logits, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size,
                                      size_class, mean_size, B, L2, num_bins, num_sizes, weights, scratch)
    losses = {"total_loss": total}
    for i, k in enumerate(LOSS_NAMES[1:], start=1):
        losses[k] = rest[i]
    return losses, (rest[8], rest[9], rest[10]), rest[11]


class IouMetrics:
    """IoU_2D / IoU_3D / IoU_>=thresh of models/det_base.py:480-503 from the row-major logits: one small HIP launch
    (csrc/box_iou.hip iou_metric_kernel) on a SIDE stream beside the loss tail -- nothing in the backward depends on it, so
    the step's critical path does not see it; join() makes the caller's stream wait for it."""

    def __init__(self):
        self.per_dev = {}
        self._join = None

    def __call__(self, logits, B, L2, cls_label, center_ref2, box3d_center, box3d_heading, box3d_size, mean_size, num_bins,
                 num_sizes, iou_thresh):
        dev = logits.device
        key = str(dev)
        if key not in self.per_dev:
            with torch.cuda.device(dev):
                self.per_dev[key] = (torch.cuda.Stream(device=dev), torch.cuda.Event(enable_timing=False),
                                     torch.cuda.Event(enable_timing=False),
                                     torch.zeros(8, dtype=torch.float32, device=dev))
        side, ev_in, ev_out, scratch = self.per_dev[key]
        if cls_label.dtype != torch.int64:
            raise TypeError("cls_label must be int64 (got %s)" % cls_label.dtype)
        lg = logits.detach()
        out = torch.empty(4, dtype=torch.float32, device=dev)
        args = [cls_label.contiguous(), center_ref2.contiguous().float(), box3d_center.contiguous().float(),
                box3d_heading.contiguous().float(), box3d_size.contiguous().float(), mean_size.contiguous().float()]
        cur = torch.cuda.current_stream(dev)
        ev_in.record(cur)
        side.wait_event(ev_in)
        with torch.cuda.stream(side):
            with torch.cuda.device(dev):
                rc = _native.lib().fcn_det_iou_metrics(lg.data_ptr(), int(lg.shape[1]), *[t.data_ptr() for t in args], int(B),
                                                       int(L2), int(num_bins), int(num_sizes), float(iou_thresh),
                                                       scratch.data_ptr(), out.data_ptr(), _native.current_stream(dev))
            _native.check(rc, "fcn_det_iou_metrics")
            ev_out.record(side)
        for t in [lg, out] + args:
            t.record_stream(side)
        self._join = (cur, ev_out)
        return out

    def join(self):
        if self._join is not None:
            cur, ev = self._join
            cur.wait_event(ev)
            self._join = None
