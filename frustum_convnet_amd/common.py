"""Layer factories, initialisers and loss helpers used on the hot path.

Mirrors the parts of the reference's models/common.py that models/det_base.py calls:
Conv1d / Conv2d / DeConv1d factories (:38-49,59-63), init_params (:18-35), softmax_focal_loss_ignore
(:217-232) and get_accuracy (:80-94).  The focal loss and accuracy are written mask-weighted (no
nonzero()/boolean indexing) so the whole forward stays free of host synchronisation and can be captured
in a hipGraph; values are identical up to fp32 summation order.
"""
import torch
import torch.nn as nn


def init_params(m, method="constant"):
    if isinstance(m, (list, tuple)):
        for im in m:
            init_params(im, method)
        return
    if method == "xavier_uniform":
        nn.init.xavier_uniform_(m.weight.data)
    elif method == "kaiming_normal":
        nn.init.kaiming_normal_(m.weight.data, mode="fan_in")
    elif isinstance(method, (int, float)):
        m.weight.data.fill_(method)
    else:
        raise ValueError("unknown method.")
    if m.bias is not None:
        m.bias.data.zero_()


def _seq(conv, norm, i_c, o_c, k, s, p, bn):
    if bn:
        return nn.Sequential(conv(i_c, o_c, k, s, p, bias=False), norm(o_c), nn.ReLU(True))
    return nn.Sequential(conv(i_c, o_c, k, s, p), nn.ReLU(True))


def Conv1d(i_c, o_c, k, s=1, p=0, bn=True):
    return _seq(nn.Conv1d, nn.BatchNorm1d, i_c, o_c, k, s, p, bn)


def Conv2d(i_c, o_c, k, s=1, p=0, bn=True):
    return _seq(nn.Conv2d, nn.BatchNorm2d, i_c, o_c, k, s, p, bn)


def DeConv1d(i_c, o_c, k, s=1, p=0, bn=True):
    return _seq(nn.ConvTranspose1d, nn.BatchNorm1d, i_c, o_c, k, s, p, bn)


def bn_momentum(bn):
    """BatchNorm momentum for the HIP kernels.  momentum=None means a cumulative moving average in PyTorch, which the
    kernels do not implement (no shipped cfg uses it): fail loudly instead of substituting a value."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm with momentum=None (cumulative average) is not supported by the HIP path")
    return float(bn.momentum)


def masked_mean(x, mask, count=None):
    """mean of x over rows where mask is 1 (x: (R,), mask: (R,) float)."""
    if count is None:
        count = mask.sum()
    # an empty mask reports 0 (as the fused HIP tail does), not 0/0
    return (x * mask).sum() / count.clamp(min=1.0)


def softmax_focal_loss_ignore(prob, target, alpha=0.25, gamma=2, ignore_idx=-1):
    keep = (target != ignore_idx).to(prob.dtype)
    num_fg = (target > 0).sum()
    tclamp = target.clamp(min=0)
    alpha_t = (1 - alpha) * (target == 0).to(prob.dtype) + alpha * (target >= 1).to(prob.dtype)
    prob_t = torch.gather(prob, 1, tclamp.view(-1, 1)).squeeze(1)
    loss = -alpha_t * (1 - prob_t) ** gamma * torch.log(prob_t + 1e-14)
    return (loss * keep).sum() / (num_fg + 1e-14)


def get_accuracy(output, target, mask=None, ignore=None):
    """Fraction of rows whose argmax equals target, over rows with mask==1 (and target != ignore)."""
    assert output.shape[0] == target.shape[0]
    m = torch.ones_like(target, dtype=output.dtype) if mask is None else mask.to(output.dtype)
    if ignore is not None:
        m = m * (target != ignore).to(output.dtype)
    pred = torch.argmax(output, -1)
    correct = ((pred.view(-1) == target.view(-1)).to(output.dtype) * m).sum()
    return correct / m.sum()
