#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference (read-only at
/root/reference) on CPU.  Runs only in the build container; nothing from the reference is
copied -- the fixtures hold inputs' seeds and the reference modules' outputs.

Stand-ins injected before importing models/det_base.py (both unbuildable here):
  ops.query_depth_point.query_depth_point  -> CPU QueryDepthPoint backed by oracle/grouping.py
                                              (reference op is CUDA-only: query_depth_point.py:23-24)
  ops.pybind11.box_ops_cc                  -> rbbox_iou_3d_pair returning zeros (boost::geometry absent;
                                              feeds no_grad metrics only, det_base.py:480-503)

Usage:  python tests/golden/make_golden.py           (rewrites the KITTI-model fixtures tests/golden/*.npz)
        python tests/golden/make_golden.py full [people] [refine] [sunrgbd]   (B = 32 fixtures of those configurations)
        python tests/golden/make_golden.py sunrgbd   (writes sunrgbd_b4_n1024.npz from models/det_base_sunrgbd.py)
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import grouping  # noqa: E402
from frustum_convnet_amd import synth  # noqa: E402


def _inject_standins():
    import yaml
    _orig = yaml.load
    yaml.load = lambda s, Loader=None: _orig(s, Loader=Loader or yaml.FullLoader)  # config.py:228 has no Loader

    class _QDPFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dis_z, nsample, xyz1, xyz2):
            idx, cnt = grouping.query_depth_point(dis_z, nsample, xyz1.detach().numpy(), xyz2.detach().numpy())
            return torch.from_numpy(idx), torch.from_numpy(cnt)

        @staticmethod
        def backward(ctx, *g):
            return (None,) * 4

    class QueryDepthPoint(torch.nn.Module):
        def __init__(self, dis_z, nsample):
            super().__init__()
            self.dis_z, self.nsample = dis_z, nsample

        def forward(self, xyz1, xyz2):
            return _QDPFn.apply(self.dis_z, self.nsample, xyz1, xyz2)

    for name in ("ops", "ops.query_depth_point", "ops.pybind11"):
        sys.modules.setdefault(name, types.ModuleType(name))
    m = types.ModuleType("ops.query_depth_point.query_depth_point")
    m.QueryDepthPoint = QueryDepthPoint
    sys.modules[m.__name__] = m
    m2 = types.ModuleType("ops.pybind11.box_ops_cc")
    m2.rbbox_iou_3d_pair = lambda a, b: np.zeros((a.shape[0], 2), dtype=np.float64)
    sys.modules[m2.__name__] = m2


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _ref_model(height_half, seed=7, module="models.det_base", dataset="KITTI", num_vec=3):
    from configs.config import cfg
    cfg.immutable(False)
    cfg.DATA.HEIGHT_HALF = tuple(height_half)
    cfg.DATA.STRIDE = tuple(height_half)
    cfg.DATA.DATASET_NAME = dataset
    import importlib
    det_base = importlib.import_module(module)
    importlib.reload(det_base)
    model = det_base.PointNetDet(3, num_vec=num_vec, num_classes=2)
    synth.fill_state_dict(model.state_dict(), seed=seed)
    return model


def _capture_feats(model):
    store = {}

    def hook(mod, inp, out):
        store["feats"] = [o.detach().clone() for o in out]
    h = model.feat_net.register_forward_hook(hook)
    return store, h


def _oracle64_grads(model, data_np, strides):
    """The fp64 referee: oracle/det_ref.py evaluated in double precision on the same weights and batch -> {name: grad}."""
    from oracle import det_ref
    sd = {k: (v.detach().clone().double() if v.dtype.is_floating_point else v.detach().clone())
          for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in synth.to_torch(data_np).items()}
    _, _, lo = det_ref.forward(sd, d64, tuple(strides), training=True)
    lo["total_loss"].backward()
    return {k: v.grad for k, v in sd.items() if v.dtype.is_floating_point and v.grad is not None}, float(lo["total_loss"])


def run_case(name, batch, npoint, strides, variant, tilt, z_range=None, full_idx=True, logit_samples=None,
             grads=True, seed=1234, sunrgbd=False, oracle64=False):
    nscale = len(strides)
    if sunrgbd:     # models/det_base_sunrgbd.py with the SUN-RGBD class table (cfgs/det_sample_sunrgbd.yaml)
        from oracle.det_ref import MEAN_SIZE_SUNRGBD, NSAMPLE_SUNRGBD
        data_np = synth.make_batch(batch, npoint, strides=strides, max_depth=8.0, seed=seed, variant=variant, tilt=tilt,
                                   z_range=z_range, num_classes=10, mean_sizes=MEAN_SIZE_SUNRGBD)
        model = _ref_model(strides, module="models.det_base_sunrgbd", dataset="SUNRGBD", num_vec=10)
        nsamples = NSAMPLE_SUNRGBD
    else:
        data_np = synth.make_batch(batch, npoint, strides=strides, seed=seed, variant=variant, tilt=tilt,
                                   z_range=z_range)
        model = _ref_model(strides)
        nsamples = (32, 64, 64, 128)
    data = synth.to_torch(data_np)
    keys = list(model.state_dict().keys())
    shapes = [tuple(v.shape) for v in model.state_dict().values()]
    out = {"meta_batch": batch, "meta_npoint": npoint, "meta_strides": np.array(strides),
           "meta_variant": variant, "meta_tilt": np.array(tilt), "meta_seed": seed,
           "meta_z_range": np.array(z_range if z_range is not None else [np.nan, np.nan]),
           "state_keys": np.array(keys), "state_shapes": np.array([str(s) for s in shapes])}

    # grouping through the stand-in (oracle), recorded per scale
    pc = data["point_cloud"][:, :3].contiguous()
    for s in range(nscale):
        idx, cnt = grouping.query_depth_point(float(strides[s]), nsamples[s], pc.numpy(),
                                              data["center_ref%d" % (s + 1)].numpy())
        i2, c2 = grouping.query_depth_point_numpy(float(strides[s]), nsamples[s], pc.numpy(),
                                                  data["center_ref%d" % (s + 1)].numpy())
        assert np.array_equal(idx, i2) and np.array_equal(cnt, c2)
        out["cnt%d" % (s + 1)] = cnt
        out["idx%d_sha" % (s + 1)] = np.array(_sha(idx))
        if full_idx:
            assert idx.max() < 32768
            out["idx%d" % (s + 1)] = idx.astype(np.int16)

    g64 = None
    if oracle64:          # BEFORE the reference's forward updates the running statistics (train mode does not read them)
        g64, loss64 = _oracle64_grads(model, data_np, strides)
        out["loss64_total"] = np.array(loss64)
    # training-mode forward + backward through the reference modules
    model.train()
    store, h = _capture_feats(model)
    caught = {}
    hk1 = model.cls_out.register_forward_hook(lambda m, i, o: caught.__setitem__("cls", o.detach().clone()))
    hk2 = model.reg_out.register_forward_hook(lambda m, i, o: caught.__setitem__("reg", o.detach().clone()))
    losses, metrics = model(data)
    feats = store["feats"]
    for s in range(nscale):
        f = feats[s].numpy()
        out["feat%d_sum" % (s + 1)] = np.array([f.astype(np.float64).sum(), np.abs(f).astype(np.float64).sum()])
        out["feat%d_b0" % (s + 1)] = f[0, ::7, :].copy()           # every 7th channel of sample 0
    sel = list(range(batch)) if logit_samples is None else list(logit_samples)
    out["logit_samples"] = np.array(sel)
    out["cls_train"] = caught["cls"][sel].numpy()
    out["reg_train"] = caught["reg"][sel].numpy()
    out["loss_names"] = np.array(list(losses.keys()))
    out["loss_train"] = np.array([float(v.detach().double()) for v in losses.values()])
    if grads:
        losses["total_loss"].backward()
        gn = []
        for k, p in model.named_parameters():
            gn.append(float(p.grad.double().norm()))
        out["grad_names"] = np.array([k for k, _ in model.named_parameters()])
        out["grad_norms"] = np.array(gn)
        if g64 is not None:     # the same norms from the fp64 oracle: how far the reference's own fp32 arithmetic sits from them
            out["grad_norms64"] = np.array([float(g64[k].norm()) for k, _ in model.named_parameters()])
        named = dict(model.named_parameters())
        for k in ("cls_out.weight", "cls_out.bias", "reg_out.weight", "reg_out.bias",
                  "feat_net.pointnet1.conv1.0.weight", "feat_net.pointnet1.conv1.1.weight",
                  "feat_net.pointnet1.conv1.1.bias", "feat_net.pointnet4.conv3.0.weight",
                  "feat_net.pointnet4.conv3.1.weight", "feat_net.pointnet2.conv2.0.weight",
                  "conv_net.block1_conv1.0.weight", "feat_net.pointnet5.conv3.0.weight",
                  "conv_net.block5_deconv.0.weight", "conv_net.block5_merge.1.weight"):
            if k not in named:
                continue
            g = named[k].grad.numpy()
            if g.size > 40000:
                g = g.reshape(g.shape[0], -1)[::8, ::4]
            out["grad::" + k] = g.copy()
            if g64 is not None:
                q = g64[k].numpy()
                if q.size > 40000:
                    q = q.reshape(q.shape[0], -1)[::8, ::4]
                out["grad64::" + k] = q.copy()
    # running stats after one step
    sd = model.state_dict()
    rs_names, rs_vals = [], []
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            rs_names.append(k)
            rs_vals.append(v.numpy().astype(np.float32).ravel())
    out["rs_names"] = np.array(rs_names)
    out["rs_sizes"] = np.array([len(v) for v in rs_vals])
    out["rs_concat"] = np.concatenate(rs_vals)

    # eval-mode forward with the updated running stats, labels dropped -> 6-tuple
    model.eval()
    ev = {k: v for k, v in data.items() if k in ("point_cloud", "one_hot", "center_ref1", "center_ref2",
                                                 "center_ref3", "center_ref4", "center_ref5")}
    with torch.no_grad():
        tup = model(ev)
    out["cls_eval"] = caught["cls"][sel].numpy()
    out["reg_eval"] = caught["reg"][sel].numpy()
    for nm, t in zip(("cls_probs", "center", "heading", "size", "heading_probs", "size_probs"), tup):
        out["eval_" + nm] = t[sel].numpy()
    h.remove(); hk1.remove(); hk2.remove()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-24s %8.1f KB   total_loss %.6f" % (name, os.path.getsize(path) / 1024.0, out["loss_train"][0]))


def testpy_case():
    """Scenario of the reference's ops/query_depth_point/test.py:10-28 (shapes (2,3,50)/(2,3,10),
    dis_z 0.2, nsample 4, queries = first 10 points) with its CPU mask criterion evaluated in torch."""
    xyz1 = (synth.uniform01(99, synth.stream_id("testpy"), (2, 3, 50)) * 2 - 1).astype(np.float32)
    xyz2 = xyz1[:, :, :10].copy()
    t1, t2 = torch.from_numpy(xyz1), torch.from_numpy(xyz2)
    mask = torch.zeros(2, 10, 50)
    for i in range(2):
        for j in range(10):
            mask[i, j] = (torch.abs(t1[i, 2] - t2[i, 2, j]) < 0.2)
    idx, cnt = grouping.query_depth_point(0.2, 4, xyz1, xyz2)
    np.savez_compressed(os.path.join(HERE, "qdp_testpy.npz"), xyz1=xyz1, xyz2=xyz2,
                        mask=mask.numpy().astype(np.uint8), idx=idx, cnt=cnt)
    print("qdp_testpy written")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _inject_standins()
    sys.path.insert(0, REF)
    if len(sys.argv) > 1 and sys.argv[1] == "sunrgbd":     # only the 5-scale fixture (the others stay as committed)
        # seed 1329: of seeds 1234..1329 the one whose ConvFeatNet pre-ReLU activations stay furthest from zero (1.3e-5 in
        # the fp64 oracle).  With seed 1234 one block3_conv1 activation sat at 1.2e-6 -- inside fp32 noise -- and the
        # side of the ReLU kink it fell on moved the gradient norms of the early layers by 1 % (reference fp32 and the
        # fp64 oracle on one side, a decomposed-BN fp32 evaluation and the HIP path on the other): a fixture must not
        # test a coin flip.
        run_case("sunrgbd_b4_n1024", 4, 1024, (0.1, 0.2, 0.4, 0.8, 1.6), "car", (0.01, 0.05), full_idx=False,
                 sunrgbd=True, seed=1329)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "full":       # full-size (B = 32) fixtures of the other three configurations (round 4)
        ppl_ = (0.1, 0.2, 0.4, 0.8)
        which = sys.argv[2:] or ["people", "refine", "sunrgbd"]
        if "people" in which:
            run_case("people_b32_n1024", 32, 1024, ppl_, "car", (0.01, 0.05), full_idx=False, logit_samples=(0, 17), oracle64=True)
        if "refine" in which:
            run_case("refine_b32_n512", 32, 512, ppl_, "uniform", (0.0, 0.0), z_range=(-1.0, 1.0), full_idx=False,
                     logit_samples=(0, 17), oracle64=True)
        if "sunrgbd" in which:
            run_case("sunrgbd_b32_n2048", 32, 2048, (0.1, 0.2, 0.4, 0.8, 1.6), "car", (0.01, 0.05), full_idx=False,
                     sunrgbd=True, seed=1329, logit_samples=(0, 17), oracle64=True)
        return
    testpy_case()
    car = (0.25, 0.5, 1.0, 2.0)
    ppl = (0.1, 0.2, 0.4, 0.8)
    run_case("car_b4_n512", 4, 512, car, "car", (0.01, 0.05))
    run_case("car_b4_n512_uniform", 4, 512, car, "uniform", (0.0, 0.0), grads=False)
    run_case("car_b32_n1024", 32, 1024, car, "car", (0.01, 0.05), full_idx=False, logit_samples=(0, 17))
    run_case("people_b2_n512", 2, 512, ppl, "car", (0.01, 0.05), full_idx=False, grads=False)
    run_case("refine_b4_n512", 4, 512, ppl, "uniform", (0.0, 0.0), z_range=(-1.0, 1.0), grads=False)


if __name__ == "__main__":
    main()
