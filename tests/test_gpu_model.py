"""-m gpu: the whole drop-in PointNetDet (HIP grouping + fused PointNet scales, MIOpen FCN, loss tail)
against golden vectors captured from the reference's own modules (tests/golden/make_golden.py).
Tolerances (north_star): idx bit-exact (test_gpu_grouping), raw cls/box logits abs 1e-4 fp32."""
import numpy as np
import pytest
import torch

from helpers import load_golden, golden_inputs, golden_state_dict
from frustum_convnet_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(g):
    from frustum_convnet_amd.config import cfg, reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    cfg.DATA.HEIGHT_HALF = tuple(float(x) for x in g["meta_strides"])
    cfg.DATA.STRIDE = cfg.DATA.HEIGHT_HALF
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    sd = golden_state_dict(g)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    return m.cuda()


@pytest.mark.parametrize("case", ["car_b4_n512", "car_b4_n512_uniform", "people_b2_n512", "refine_b4_n512",
                                  "car_b32_n1024"])
def test_train_eval_parity(case):
    g = load_golden(case)
    data = synth.to_torch(golden_inputs(g), "cuda")
    m = _model(g)
    m.train()
    losses, metrics = m(data)
    cls, reg = m.last_logits
    sel = torch.as_tensor(g["logit_samples"]).cuda()
    d_cls = np.abs(cls[sel].detach().cpu().numpy() - g["cls_train"]).max()
    d_reg = np.abs(reg[sel].detach().cpu().numpy() - g["reg_train"]).max()
    print(case, "train logits max abs diff: cls %.3e reg %.3e" % (d_cls, d_reg))
    assert d_cls < TOL and d_reg < TOL
    for nm, ref in zip(g["loss_names"], g["loss_train"]):
        got = float(losses[str(nm)])
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), (nm, got, ref)
    if "grad_norms" in g.files:
        losses["total_loss"].backward()
        named = dict(m.named_parameters())
        worst = 0.0
        for nm, ref in zip(g["grad_names"], g["grad_norms"]):
            got = float(named[str(nm)].grad.double().norm())
            worst = max(worst, abs(got - ref) / max(ref, 1e-3))
            assert abs(got - ref) <= 1e-3 * max(ref, 1e-3), (nm, got, ref)
        print(case, "worst relative grad-norm diff %.3e" % worst)
        for k in g.files:
            if k.startswith("grad::"):
                gr = named[k[6:]].grad.detach().cpu().numpy()
                if gr.size > 40000:
                    gr = gr.reshape(gr.shape[0], -1)[::8, ::4]
                ref = g[k]
                assert np.abs(gr - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, k
    sd = m.state_dict()
    off = 0
    for nm, n in zip(g["rs_names"], g["rs_sizes"]):
        ref = g["rs_concat"][off:off + n]
        off += n
        assert np.allclose(sd[str(nm)].cpu().numpy(), ref, rtol=1e-4, atol=1e-5), nm
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1, k
    # eval branch: 6-tuple from running statistics
    m.eval()
    ev = {k: v for k, v in data.items() if k in ("point_cloud", "one_hot", "center_ref1", "center_ref2",
                                                 "center_ref3", "center_ref4")}
    with torch.no_grad():
        tup = m(ev)
    cls, reg = m.last_logits
    assert np.abs(cls[sel].cpu().numpy() - g["cls_eval"]).max() < TOL
    assert np.abs(reg[sel].cpu().numpy() - g["reg_eval"]).max() < TOL
    names = ("cls_probs", "center", "heading", "size", "heading_probs", "size_probs")
    for nm, t in zip(names, tup):
        ref = g["eval_" + nm]
        got = t[sel].cpu().numpy()
        assert got.shape == ref.shape, nm
        if nm in ("heading", "size"):
            # argmax-dependent: compare where the deciding top-2 probability gap is clear
            hp, sp = g["eval_heading_probs"], g["eval_size_probs"]
            gap = lambda p: np.sort(p, -1)[..., -1] - np.sort(p, -1)[..., -2]
            ok = (gap(hp) > 1e-3) & (gap(sp) > 1e-3)
            assert np.abs(got - ref)[ok].max() < 1e-3, nm
        else:
            assert np.abs(got - ref).max() < TOL, nm


def test_dense_module_api_matches_oracle():
    """PointNetModule.forward keeps the reference's (B, C3, L, nsample) masked return."""
    from oracle import det_ref
    g = load_golden("car_b4_n512")
    data_np = golden_inputs(g)
    m = _model(g)
    m.train()
    pc = torch.from_numpy(data_np["point_cloud"]).cuda()
    ref = torch.from_numpy(data_np["center_ref3"]).cuda()
    out = m.feat_net.pointnet3(pc, None, ref)
    sd = golden_state_dict(g)
    exp, _, _ = det_ref.pointnet_module(torch.from_numpy(data_np["point_cloud"]), torch.from_numpy(data_np["center_ref3"]),
                                        sd, "feat_net.pointnet3", 1.0, 64, True)
    assert out.shape == exp.shape
    assert (out.cpu() - exp).abs().max() < 2e-4


def test_cpu_input_fails_loudly():
    g = load_golden("car_b4_n512")
    m = _model(g).cpu()
    data = synth.to_torch(golden_inputs(g), "cpu")
    with pytest.raises((RuntimeError, AssertionError)):
        m(data)


def test_two_forwards_before_backward_do_not_share_workspace():
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    m = _model(g)
    m.train()
    l1, _ = m(data)
    l2, _ = m(data)           # second forward before the first backward
    l1["total_loss"].backward()
    g1 = m.feat_net.pointnet4.conv3[0].weight.grad.clone()
    m.zero_grad()
    l2["total_loss"].backward()
    # second graph saw updated running stats but identical batch statistics -> same gradients
    g2 = m.feat_net.pointnet4.conv3[0].weight.grad
    assert float((g1 - g2).abs().max()) <= 1e-4 * float(g1.abs().max())


def test_fused_loss_tail_matches_torch_tail():
    """C-ABI fcn_det_loss_tail (one launch) vs the mask-weighted torch tail and the oracle's tail: values and
    d(total)/d(logits)."""
    from oracle import det_ref
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    m = _model(g)
    m.train()
    outs = {}
    for fused in (True, False):
        m.fused_loss = fused
        m.zero_grad()
        losses, metrics = m(data)
        cls, reg = m.last_logits
        cls.retain_grad(); reg.retain_grad()
        losses["total_loss"].backward()
        outs[fused] = ({k: float(v) for k, v in losses.items()}, {k: float(v) for k, v in metrics.items()},
                       cls.grad.clone(), reg.grad.clone())
    for k, v in outs[False][0].items():
        assert abs(outs[True][0][k] - v) <= 2e-5 * max(1.0, abs(v)), k
    for k in ("cls_acc", "head_acc", "size_acc"):
        assert abs(outs[True][1][k] - outs[False][1][k]) < 1e-6, k
    for i in (2, 3):
        a, b = outs[True][i], outs[False][i]
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7
    cpu = synth.to_torch(golden_inputs(g))
    ref = det_ref.loss_tail(torch.from_numpy(g["cls_train"]), torch.from_numpy(g["reg_train"]), cpu)
    for k, v in ref.items():
        assert abs(outs[True][0][k] - float(v)) <= 1e-4 * max(1.0, abs(float(v))), k
