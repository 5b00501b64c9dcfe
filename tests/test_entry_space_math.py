"""CPU proof that the entry-space (dedup + multiplicity weight) algorithm the kernels implement
equals the reference's dense dataflow (oracle/det_ref.py), forward and backward."""
import numpy as np
import pytest
import torch

from oracle import det_ref, grouping
import entry_ref
from frustum_convnet_amd import synth


def _setup(B, N, L, K, mlp, dist, seed):
    d = synth.make_batch(B, N, strides=(70.0 / L,) * 4, seed=seed, variant="car", tilt=(0.01, 0.05))
    pc = torch.from_numpy(d["point_cloud"])
    ref = torch.from_numpy(d["center_ref1"])
    sd = {}
    cin = 3
    for j, co in enumerate(mlp):
        p = "m.conv%d" % (j + 1)
        sd[p + ".0.weight"] = torch.zeros(co, cin, 1, 1)
        sd[p + ".1.weight"] = torch.zeros(co)
        sd[p + ".1.bias"] = torch.zeros(co)
        sd[p + ".1.running_mean"] = torch.zeros(co)
        sd[p + ".1.running_var"] = torch.ones(co)
        cin = co
    synth.fill_state_dict(sd, seed=seed)
    return pc, ref, sd


@pytest.mark.parametrize("B,N,L,K,dist", [(2, 128, 20, 16, 1.0), (3, 64, 9, 8, 2.0), (1, 200, 33, 4, 0.7)])
def test_forward_backward_equivalence(B, N, L, K, dist):
    mlp = (16, 24, 40)
    pc, ref, sd = _setup(B, N, L, K, mlp, dist, seed=5)
    for k, v in sd.items():
        if "running" not in k:
            v.requires_grad_(True)
    rec = det_ref.BNState()
    g, idx, cnt = det_ref.pointnet_module(pc, ref, sd, "m", dist, K, True, rec)
    feat = g.max(-1)[0]
    dfeat = torch.from_numpy(synth.normalish(3, 1, tuple(feat.shape)).astype(np.float32))
    (feat * dfeat).sum().backward()
    assert (cnt == 0).any()                # the cases are chosen to contain empty windows
    assert (cnt == K).any()                # ... and saturated ones

    c = entry_ref.compact(idx, cnt, pc, ref, K)
    W = [sd["m.conv%d.0.weight" % j].detach().view(mlp[j - 1], -1) for j in (1, 2, 3)]
    G = [sd["m.conv%d.1.weight" % j].detach() for j in (1, 2, 3)]
    Bt = [sd["m.conv%d.1.bias" % j].detach() for j in (1, 2, 3)]
    f = entry_ref.forward(c, W[0], G[0], Bt[0], W[1], G[1], Bt[1], W[2], G[2], Bt[2])
    assert torch.allclose(f["feat"], feat.detach(), atol=2e-5, rtol=1e-5)
    for j in range(3):
        mean, var, n = rec.stats["m.conv%d.1" % (j + 1)]
        assert n == B * L * K
        assert torch.allclose(f["mean"][j].float(), mean, atol=1e-5, rtol=1e-5)
        assert torch.allclose(f["var"][j].float(), var, atol=1e-5, rtol=1e-4)
    r = entry_ref.backward(c, f, dfeat, W[0], G[0], W[1], G[1], W[2], G[2])
    for j in (1, 2, 3):
        gw = sd["m.conv%d.0.weight" % j].grad.view(mlp[j - 1], -1)
        sc = gw.abs().max()
        assert (r["dW%d" % j] - gw).abs().max() <= 2e-4 * sc + 1e-6, j
        gg = sd["m.conv%d.1.weight" % j].grad
        gb = sd["m.conv%d.1.bias" % j].grad
        assert (r["dg%d" % j] - gg).abs().max() <= 2e-4 * gg.abs().max() + 1e-6
        assert (r["db%d" % j] - gb).abs().max() <= 2e-4 * gb.abs().max() + 1e-6
