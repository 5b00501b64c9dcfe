"""ORACLE -- test infrastructure only (see oracle/__init__.py).

Plain PyTorch-CPU functional restatement of the reference's per-frustum hot path
(models/det_base.py): grouping -> gather -> 3 x [1x1 conv, BN, ReLU] -> mask -> max over K
-> one-hot concat -> Conv1d FCN -> heads -> loss tail.  Dense dataflow, exactly as the
reference materialises it (every (B,C,L,K) tensor exists), no fusion, no dedup.

Parameters come from a state_dict with the reference's key names
(feat_net.pointnet{1-4}.conv{1-3}.{0,1}.*, conv_net.block*.{0,1}.*, cls_out.*, reg_out.*).
Pinned against fixtures generated from the reference modules by tests/golden/make_golden.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import grouping

NSAMPLE = (32, 64, 64, 128)           # models/det_base.py:114-124
MEAN_SIZE = np.array([[3.88311640418, 1.62856739989, 1.52563191462],
                      [0.84422524, 0.66068622, 1.76255119],
                      [1.76282397, 0.59706367, 1.73698127]])  # datasets/dataset_info.py:6-10
NSAMPLE_SUNRGBD = (128, 128, 256, 256, 256)   # models/det_base_sunrgbd.py:113-128
MEAN_SIZE_SUNRGBD = np.array([[0.765840, 1.398258, 0.472728], [2.114256, 1.620300, 0.927272],
                              [0.404671, 1.071108, 1.688889], [0.591958, 0.552978, 0.827272],
                              [0.695190, 1.346299, 0.736364], [0.528526, 1.002642, 1.172878],
                              [0.500618, 0.632163, 0.683424], [0.923508, 1.867419, 0.845495],
                              [0.791118, 1.279516, 0.718182], [0.699104, 0.454178, 0.756250]])  # datasets/dataset_info.py:24-35
LOSS_W = dict(BOX=1.0, CORNER=10.0, HEAD_REG=20.0, SIZE_REG=20.0)  # configs/config.py:161-167


class BNState:
    """Collects the batch statistics each BN layer saw (for running-stat checks)."""

    def __init__(self):
        self.stats = {}


def _bn(x, sd, prefix, training, rec=None):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if rec is not None:
            n = x.numel() // x.shape[1]
            rec.stats[prefix] = (mean.detach().clone(), var.detach().clone(), n)
    else:
        mean, var = rm, rv
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + 1e-5) * w.view(shape) + b.view(shape)


def _cbr2d(x, sd, prefix, training, rec):
    # models/common.py:45-49 : Conv2d(k=1, bias=False) + BatchNorm2d + ReLU
    y = F.conv2d(x, sd[prefix + ".0.weight"])
    return torch.relu(_bn(y, sd, prefix + ".1", training, rec))


def _cbr1d(x, sd, prefix, training, rec, stride=1, pad=0):
    y = F.conv1d(x, sd[prefix + ".0.weight"], stride=stride, padding=pad)
    return torch.relu(_bn(y, sd, prefix + ".1", training, rec))


def _dbr1d(x, sd, prefix, training, rec, stride):
    y = F.conv_transpose1d(x, sd[prefix + ".0.weight"], stride=stride)
    return torch.relu(_bn(y, sd, prefix + ".1", training, rec))


def pointnet_module(pc, ref, sd, prefix, dist, nsample, training, rec=None, group=None):
    """models/det_base.py:62-103.  pc (B,3,N), ref (B,3,L) -> masked (B,C3,L,K), idx, cnt."""
    B, _, N = pc.shape
    L = ref.shape[2]
    if group is None:
        idx_np, cnt_np = grouping.query_depth_point(dist, nsample, pc.detach().numpy(), ref.detach().numpy())
    else:
        idx_np, cnt_np = group
    idx = torch.from_numpy(idx_np)
    cnt = torch.from_numpy(cnt_np)
    g = torch.gather(pc, 2, idx.view(B, 1, L * nsample).expand(-1, 3, -1)).view(B, 3, L, nsample)
    g = g - ref.unsqueeze(3)
    g = _cbr2d(g, sd, prefix + ".conv1", training, rec)
    g = _cbr2d(g, sd, prefix + ".conv2", training, rec)
    g = _cbr2d(g, sd, prefix + ".conv3", training, rec)
    valid = (cnt > 0).view(B, 1, L, 1).to(g.dtype)
    return g * valid, idx, cnt


def pointnet_feat(pc, refs, one_hot, sd, height_half, training, rec=None, keep=None, nsample=None):
    """models/det_base.py:126-159 -> four pooled (B,C+3,L_s) maps (five with models/det_base_sunrgbd.py:130-170)."""
    feats = []
    if nsample is None:
        nsample = NSAMPLE if len(refs) == 4 else NSAMPLE_SUNRGBD
    for s in range(len(refs)):
        g, idx, cnt = pointnet_module(pc, refs[s], sd, "feat_net.pointnet%d" % (s + 1),
                                      float(height_half[s]), nsample[s], training, rec)
        f = g.max(dim=-1)[0]
        if keep is not None:
            keep["idx%d" % (s + 1)] = idx
            keep["cnt%d" % (s + 1)] = cnt
            keep["pooled%d" % (s + 1)] = f
        if one_hot is not None:
            f = torch.cat([f, one_hot.unsqueeze(-1).expand(-1, -1, f.shape[-1])], 1)
        feats.append(f)
    return feats


def conv_feat_net(*feats, sd, training, rec=None, p="conv_net"):
    """models/det_base.py:196-224 (four maps) / models/det_base_sunrgbd.py:213-251 (five): block1_conv1, then per level j
    >= 2 the stride-2 conv1, conv2 and the 1x1 merge with map j; every merge output is upsampled by block{j}_deconv
    (kernel = stride = 2^(j-2)), cut to the length of the first and concatenated."""
    x = _cbr1d(feats[0], sd, p + ".block1_conv1", training, rec, 1, 1)
    ups = []
    for j in range(2, len(feats) + 1):
        x = _cbr1d(x, sd, p + ".block%d_conv1" % j, training, rec, 2, 1)
        x = _cbr1d(x, sd, p + ".block%d_conv2" % j, training, rec, 1, 1)
        x = _cbr1d(torch.cat([x, feats[j - 1]], 1), sd, p + ".block%d_merge" % j, training, rec)
        ups.append((j, x))
    ups = [_dbr1d(xx, sd, p + ".block%d_deconv" % j, training, rec, 1 << (j - 2)) for j, xx in ups]
    Lk = ups[0].shape[-1]
    return torch.cat([ups[0]] + [u[:, :, :Lk] for u in ups[1:]], 1)


def heads(x, sd):
    """models/det_base.py:367-368 -> raw (B,2,L2), (B,39,L2)."""
    cls = F.conv1d(x, sd["cls_out.weight"], sd["cls_out.bias"])
    reg = F.conv1d(x, sd["reg_out.weight"], sd["reg_out.bias"])
    return cls, reg


# ---- loss tail (models/det_base.py:414-476, models/common.py:217-232, model_util.py, box_transform.py)

def huber(err, delta):
    a = err.abs()
    q = torch.clamp(a, max=delta)
    return (0.5 * q * q + delta * (a - q)).mean()


def box_corners(c, h, s):
    l, w, hh = s[:, 0], s[:, 1], s[:, 2]
    xs = torch.stack([l, l, -l, -l, l, l, -l, -l], 1) / 2
    ys = torch.stack([hh, hh, hh, hh, -hh, -hh, -hh, -hh], 1) / 2
    zs = torch.stack([w, -w, -w, w, w, -w, -w, w], 1) / 2
    co, si = torch.cos(h), torch.sin(h)
    x = co[:, None] * xs + si[:, None] * zs
    z = -si[:, None] * xs + co[:, None] * zs
    return torch.stack([x, ys, z], 2) + c[:, None, :]          # (N,8,3)


def angle_encode(a, nb=12):
    a = a % (2 * np.pi)
    per = 2 * np.pi / nb
    sh = (a + per / 2) % (2 * np.pi)
    cid = torch.floor(sh / per).long()
    res = sh - (cid.to(a.dtype) * per + per / 2)
    return cid, res / (per / 2)


def angle_decode(res, cid, nb=12):
    per = 2 * np.pi / nb
    ang = cid.to(res.dtype) * per + torch.gather(res, 1, cid.unsqueeze(1)).squeeze(1) * (per / 2)
    return torch.where(ang > np.pi, ang - 2 * np.pi, ang)


def size_decode(off, mean_size, cid):
    sel = torch.gather(off, 1, cid.view(-1, 1, 1).expand(-1, -1, 3)).squeeze(1)
    ex = mean_size[cid]
    return sel * ex + ex


def loss_tail(cls_raw, reg_raw, data, nb=12, ncls=None, mean_size=None):
    """Returns dict of the 8 loss scalars of models/det_base.py:505-514 (ncls / mean_size default to the dataset the head
    width belongs to: 39 columns KITTI, 67 SUN-RGBD)."""
    if ncls is None:
        ncls = (reg_raw.shape[1] - 3 - 2 * nb) // 4
    if mean_size is None:
        mean_size = MEAN_SIZE if ncls == 3 else MEAN_SIZE_SUNRGBD
    B, _, L2 = cls_raw.shape
    cls = cls_raw.permute(0, 2, 1).reshape(-1, 2)
    out = reg_raw.permute(0, 2, 1).reshape(-1, reg_raw.shape[1])
    ref2 = data["center_ref2"].permute(0, 2, 1).reshape(-1, 3)
    mean_size = torch.from_numpy(np.asarray(mean_size)).to(cls.dtype)
    prob = F.softmax(cls, -1)
    lab = data["cls_label"].view(-1)
    fg = (lab == 1).nonzero().view(-1)
    # focal loss with ignore (common.py:217-232)
    keep = (lab != -1).nonzero().view(-1)
    nfg = (lab > 0).sum()
    t = lab[keep]
    p = prob[keep]
    alpha = 0.75 * (t == 0).to(p.dtype) + 0.25 * (t >= 1).to(p.dtype)
    pt = p[torch.arange(len(t)), t]
    cls_loss = (-alpha * (1 - pt) ** 2 * torch.log(pt + 1e-14)).sum() / (nfg + 1e-14)

    o = out[fg]
    r2 = ref2[fg]
    center, hs, hr = o[:, 0:3], o[:, 3:3 + nb], o[:, 3 + nb:3 + 2 * nb]
    ss = o[:, 3 + 2 * nb:3 + 2 * nb + ncls]
    sr = o[:, 3 + 2 * nb + ncls:].reshape(-1, ncls, 3)
    c_lab = data["box3d_center"].unsqueeze(1).expand(-1, L2, -1).reshape(-1, 3)[fg]
    h_lab = data["box3d_heading"].expand(-1, L2).reshape(-1)[fg]
    s_lab = data["box3d_size"].unsqueeze(1).expand(-1, L2, -1).reshape(-1, 3)[fg]
    sc_lab = data["size_class"].expand(-1, L2).reshape(-1)[fg]

    center_loss = huber(torch.norm(c_lab - r2 - center, 2, dim=-1), 3.0)
    hc, hres = angle_encode(h_lab, nb)
    head_cls = F.cross_entropy(hs, hc)
    head_res = huber(torch.gather(hr, 1, hc.view(-1, 1)).squeeze(1) - hres, 1.0)
    size_cls = F.cross_entropy(ss, sc_lab)
    ex = mean_size[sc_lab]
    s_res_lab = (s_lab - ex) / ex
    s_sel = torch.gather(sr, 1, sc_lab.view(-1, 1, 1).expand(-1, 1, 3)).squeeze(1)
    size_res = huber(torch.norm(s_res_lab - s_sel, 2, dim=-1), 1.0)

    cpred = r2 + center
    heading = angle_decode(hr, hc, nb)
    size = size_decode(sr, mean_size, sc_lab)
    cg = box_corners(c_lab, h_lab, s_lab)
    cgf = box_corners(c_lab, h_lab + np.pi, s_lab)
    cp = box_corners(cpred, heading, size)
    cd = torch.min(torch.norm(cp - cg, 2, dim=-1).mean(-1), torch.norm(cp - cgf, 2, dim=-1).mean(-1))
    corners = huber(cd, 1.0)

    total = cls_loss + LOSS_W["BOX"] * (center_loss + head_cls + size_cls + LOSS_W["HEAD_REG"] * head_res
                                        + LOSS_W["SIZE_REG"] * size_res + LOSS_W["CORNER"] * corners)
    return {"total_loss": total, "cls_loss": cls_loss, "center_loss": center_loss,
            "head_cls_loss": head_cls, "head_res_loss": head_res, "size_cls_loss": size_cls,
            "size_res_loss": size_res, "corners_loss": corners}


def forward(sd, data, height_half=(0.25, 0.5, 1.0, 2.0), training=True, rec=None, keep=None, with_loss=True):
    """Whole path.  Returns (cls_raw (B,2,L2), reg_raw (B,39|67,L2), losses or None).  Four or five scales by the
    center_ref keys present (and len(height_half))."""
    pc = data["point_cloud"][:, :3, :].contiguous()
    refs = [data["center_ref%d" % i] for i in range(1, 6) if ("center_ref%d" % i) in data]
    feats = pointnet_feat(pc, refs, data.get("one_hot"), sd, height_half, training, rec, keep)
    x = conv_feat_net(*feats, sd=sd, training=training, rec=rec)
    cls_raw, reg_raw = heads(x, sd)
    if keep is not None:
        keep["fcn"] = x
    losses = loss_tail(cls_raw, reg_raw, data) if (with_loss and "cls_label" in data) else None
    return cls_raw, reg_raw, losses


def updated_running_stats(sd, rec, momentum=0.1):
    """Running stats after one training forward (PyTorch BN semantics: unbiased var)."""
    out = {}
    for prefix, (mean, var, n) in rec.stats.items():
        out[prefix + ".running_mean"] = (1 - momentum) * sd[prefix + ".running_mean"] + momentum * mean
        out[prefix + ".running_var"] = (1 - momentum) * sd[prefix + ".running_var"] + momentum * var * (n / (n - 1))
    return out
