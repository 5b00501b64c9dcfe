"""Box parameter encode/decode and the small loss helpers of the train tail.

Restates the functions models/det_base.py calls from the reference's models/box_transform.py:5-65
(size/center/angle encode+decode) and models/model_util.py:9-19,48-72 (huber_loss, box corners), with the
data-dependent indexing (`angle[flag] = ...`) and host asserts replaced by torch.where so nothing
synchronises with the host.
"""
import numpy as np
import torch


def size_decode(offset, class_mean_size, size_class_label):
    sel = torch.gather(offset, 1, size_class_label.view(-1, 1, 1).expand(-1, -1, 3)).squeeze(1)
    ex = class_mean_size[size_class_label]
    return sel * ex + ex


def size_encode(gt, class_mean_size, size_class_label):
    ex = class_mean_size[size_class_label]
    return (gt - ex) / ex


def center_decode(ex, offset):
    return ex + offset


def center_encode(gt, ex):
    return gt - ex


def angle_decode(ex_res, ex_class_id, num_bins=12, to_label_format=True):
    sel = torch.gather(ex_res, 1, ex_class_id.unsqueeze(1)).squeeze(1)
    per = 2 * np.pi / float(num_bins)
    angle = ex_class_id.float() * per + sel * (per / 2)
    if to_label_format:
        angle = torch.where(angle > np.pi, angle - 2 * np.pi, angle)
    return angle


def angle_encode(gt_angle, num_bins=12):
    gt_angle = gt_angle % (2 * np.pi)
    per = 2 * np.pi / float(num_bins)
    shifted = (gt_angle + per / 2) % (2 * np.pi)
    cls_id = torch.floor(shifted / per).long()
    res = shifted - (cls_id.float() * per + per / 2)
    return cls_id, res / (per / 2)


def huber_elem(error, delta):
    """Element-wise Huber; the reference's huber_loss is the mean of this."""
    a = torch.abs(error)
    q = torch.clamp(a, max=delta)
    return 0.5 * q * q + delta * (a - q)


def huber_loss(error, delta, weight=None):
    losses = huber_elem(error, delta)
    if weight is not None:
        losses = losses * weight
    return losses.mean()


_SIGNS = {}


def _corner_signs(like):
    """(3,8) half-extent signs, cached per device/dtype (created outside any graph capture)."""
    key = (like.device, like.dtype)
    if key not in _SIGNS:
        _SIGNS[key] = 0.5 * torch.tensor([[1, 1, -1, -1, 1, 1, -1, -1],
                                          [1, 1, 1, 1, -1, -1, -1, -1],
                                          [1, -1, -1, 1, 1, -1, -1, 1]], dtype=like.dtype, device=like.device)
    return _SIGNS[key]


def get_box3d_corners_helper(centers, headings, sizes):
    """(N,3),(N,),(N,3) -> (N,8,3): corners = R_y(heading) . [+-l/2, +-h/2, +-w/2] + centre."""
    l, w, h = sizes[:, 0:1], sizes[:, 1:2], sizes[:, 2:3]
    sx, sy, sz = _corner_signs(centers)
    x, y, z = l * sx, h * sy, w * sz                                 # (N,8) each
    c, s = torch.cos(headings).unsqueeze(1), torch.sin(headings).unsqueeze(1)
    xr = c * x + s * z
    zr = -s * x + c * z
    return torch.stack([xr, y, zr], 2) + centers.unsqueeze(1)
