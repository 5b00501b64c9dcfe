"""Inference tail on the device: decode of the head outputs into label-format boxes and the
per-(frame, class) rotated 3-D NMS, plus the paired rotated IoU (row f-3) as a standalone op.

Reference: train/test_net_det.py:254-293 (numpy decode loop), :126-152 (write_detection_results_nms ->
ops/pybind11/rbbox_iou.py:294-311 rotate_nms_3d_cc -> nms_cpu.h:148-240), ops/pybind11/rbbox_iou.py:191-202
(rbbox_iou_3d_pair -> box_ops.h:173-260).  The reference copies every head output to the host and loops in numpy with
boost polygon clipping; here the three steps are HIP kernels (csrc/box_iou.hip) and only the keep lists leave the device.
No CPU fallback: CPU tensors raise.
"""
import ctypes

import numpy as np
import torch

from . import _native


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("frustum_convnet_amd.detect.%s runs on an MI355X only (got a %s tensor); there is no CPU "
                           "fallback" % (what, t.device))


def box3d_iou_pair(corners1, corners2):
    """(n,8,3) x 2 float32 device tensors (corner order of get_box3d_corners_helper) -> (n,2) [BEV IoU, 3-D IoU]."""
    _need_cuda(corners1, "box3d_iou_pair")
    assert corners1.shape == corners2.shape and corners1.shape[1:] == (8, 3)
    c1, c2 = corners1.contiguous().float(), corners2.contiguous().float()
    n = c1.shape[0]
    out = torch.zeros((n, 2), dtype=torch.float32, device=c1.device)
    with torch.cuda.device(c1.device):
        _native.check(_native.lib().fcn_box3d_iou_pair_f32(c1.data_ptr(), c2.data_ptr(), n, out.data_ptr(),
                                                           _native.current_stream(c1.device)), "fcn_box3d_iou_pair_f32")
    return out


def decode_detections(logits, center_ref2, mean_size, rot_angle, ref_center=None, rgb_prob=None, num_bins=12,
                      num_sizes=3, method="nms"):
    """logits (B*L2, ld) row-major head outputs (cols 0..1 cls, 2.. reg), center_ref2 (B,3,L2), rot_angle (B,) ->
    dets (B*L2, 8) [tx,ty,tz,l,w,h,ry,score] in label format and valid (B*L2,) int32."""
    _need_cuda(logits, "decode_detections")
    B, _, L2 = center_ref2.shape
    lg = logits.detach().contiguous().float()
    assert lg.shape[0] == B * L2
    dev = lg.device
    # every optional tensor may arrive as the reference's test loader yields it -- a CPU tensor (train/test_net_det.py:201-214)
    # -- and is moved to the logits' device here: a host pointer handed to the kernel faults the GPU
    def f(t, shape, what):
        if t is None:
            return None
        t = t.detach().to(device=dev, dtype=torch.float32)
        if t.numel() != int(np.prod(shape)):
            raise ValueError("decode_detections: %s has %d elements, expected shape %s" % (what, t.numel(), tuple(shape)))
        return t.reshape(-1).contiguous()
    if tuple(center_ref2.shape[:2]) != (B, 3):
        raise ValueError("decode_detections: center_ref2 must be (B,3,L2), got %s" % (tuple(center_ref2.shape),))
    ref2 = center_ref2.detach().to(device=dev, dtype=torch.float32).contiguous()
    ms, rot = f(mean_size, (num_sizes, 3), "mean_size"), f(rot_angle, (B,), "rot_angle")
    rc, rgb = f(ref_center, (B, 3), "ref_center"), f(rgb_prob, (B,), "rgb_prob")
    dets = torch.empty((B * L2, 8), dtype=torch.float32, device=dev)
    valid = torch.empty((B * L2,), dtype=torch.int32, device=dev)
    p = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        rc_ = _native.lib().fcn_decode_detections(lg.data_ptr(), int(lg.shape[1]), ref2.data_ptr(), ms.data_ptr(), rot.data_ptr(),
                                                  p(rc), p(rgb), int(B), int(L2), int(num_bins), int(num_sizes),
                                                  1 if method == "nms" else 0, dets.data_ptr(), valid.data_ptr(),
                                                  _native.current_stream(dev))
    _native.check(rc_, "fcn_decode_detections")
    return dets, valid


def rotate_nms_3d(dets, valid, unit_group, rows_per_unit, num_groups, thresh, top_k=300):
    """Per-group greedy rotated 3-D NMS.  dets (n,8) [cx,cy,cz,l,w,h,ry,score]; valid (n,) int32 or None; unit_group
    (n / rows_per_unit,) int32 group id of each unit (frustum).  -> keep (num_groups, top_k) int32 row indices in keep
    order, keep_cnt (num_groups,) int32 (-1: more than 4096 candidates in that group)."""
    _need_cuda(dets, "rotate_nms_3d")
    d = dets.detach().contiguous().float()
    n = d.shape[0]
    assert d.shape[1] == 8 and n % rows_per_unit == 0
    ug = unit_group.to(device=d.device, dtype=torch.int32).contiguous()
    assert ug.numel() == n // rows_per_unit
    v = None if valid is None else valid.to(device=d.device, dtype=torch.int32).contiguous()
    keep = torch.full((num_groups, top_k), -1, dtype=torch.int32, device=d.device)
    cnt = torch.zeros((num_groups,), dtype=torch.int32, device=d.device)
    with torch.cuda.device(d.device):
        rc = _native.lib().fcn_rotate_nms_3d(d.data_ptr(), None if v is None else v.data_ptr(), ug.data_ptr(),
                                             int(ug.numel()), int(rows_per_unit), int(num_groups), float(thresh), int(top_k),
                                             keep.data_ptr(), cnt.data_ptr(), _native.current_stream(d.device))
    _native.check(rc, "fcn_rotate_nms_3d")
    return keep, cnt
