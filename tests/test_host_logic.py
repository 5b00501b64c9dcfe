"""CPU tier: C-ABI library loads and exports everything include/fcn_hip.h declares (no compute calls),
the drop-in module surface (constructors, state_dict keys/shapes), config loading, loss-tail parity with the
oracle on CPU tensors, and loud failure of the hot path without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import load_golden, golden_inputs, golden_state_dict
from frustum_convnet_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from frustum_convnet_amd import _native
    from frustum_convnet_amd import build as fb
    fb.build(verbose=False)
    lib = _native.lib()
    syms = ge.declared_symbols()
    assert set(syms) == set(_native.EXPORTS)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.fcn_arch() == 950
    assert lib.fcn_pn_wgrad_rows() > 0


def test_header_cites_reference_interfaces():
    txt = open(os.path.join(ROOT, "include", "fcn_hip.h")).read()
    assert "query_depth_point_cuda.cpp:25-50" in txt and "query_depth_point_cuda_kernel.cu:16-86" in txt
    assert "models/det_base.py:75-101" in txt


def test_bad_arguments_return_codes_without_gpu():
    """Argument validation happens before any HIP call, so it is checkable on CPU."""
    import ctypes
    from frustum_convnet_amd import _native
    lib = _native.lib()
    rc = lib.fcn_query_depth_point_f32(None, 1, 0, None, 1, 0, -1, 4, 4, 0.5, 4, None, None, None)
    assert rc == 10001
    rc = lib.fcn_query_depth_point_f32(None, 1, 0, None, 1, 0, 0, 4, 4, 0.5, 4, None, None, None)
    assert rc == 0            # empty batch: nothing to do
    d = _native.PnDesc(2, 16, 4, 4, 60, 64, 128, 0, 1, 1e-5, 0.1)
    assert lib.fcn_pn_forward(ctypes.byref(d), None, None, None, None, None, None) == 10001
    # every other entry point rejects null / inconsistent arguments before touching the device
    cd = _native.CnDesc(2, (ctypes.c_int32 * 5)(280, 140, 70, 35), 3, 39, 1, 1e-5, 0.1, 0)
    sizes = (ctypes.c_int64 * 6)()
    assert lib.fcn_convnet_sizes(ctypes.byref(cd), ctypes.byref(sizes)) == 0 and all(int(v) > 0 for v in sizes)
    bad = _native.CnDesc(2, (ctypes.c_int32 * 5)(280, 141, 70, 35), 3, 39, 1, 1e-5, 0.1, 0)     # L2 != conv_len(L1)
    assert lib.fcn_convnet_sizes(ctypes.byref(bad), ctypes.byref(sizes)) == 10001
    assert lib.fcn_convnet_sizes(None, None) == 10001
    assert lib.fcn_convnet_logits_ld(ctypes.byref(cd)) == 64
    # the FCN kernels use 32-bit offsets and float-reciprocal row divisions: a batch whose B * L reaches 2^23 rows
    # (or whose arenas reach 2^30 elements: 32-bit byte offsets) is refused with FCN_E_LIMIT instead of computing wrong addresses
    ok_big = _native.CnDesc(4096, (ctypes.c_int32 * 5)(280, 140, 70, 35), 3, 39, 1, 1e-5, 0.1, 0)      # 1.1 M rows
    assert lib.fcn_convnet_sizes(ctypes.byref(ok_big), ctypes.byref(sizes)) == 0
    too_big = _native.CnDesc(32768, (ctypes.c_int32 * 5)(280, 140, 70, 35), 3, 39, 1, 1e-5, 0.1, 0)    # 9.2 M rows
    assert lib.fcn_convnet_sizes(ctypes.byref(too_big), ctypes.byref(sizes)) == 10002
    # the 5-level plan of models/det_base_sunrgbd.py: L = 80..5, block1 64 wide, 67 regression columns -> 128-wide rows
    c5 = _native.CnDesc(2, (ctypes.c_int32 * 5)(80, 40, 20, 10, 5), 10, 67, 1, 1e-5, 0.1, 0, 0, 5, 64)
    assert lib.fcn_convnet_sizes(ctypes.byref(c5), ctypes.byref(sizes)) == 0 and all(int(v) > 0 for v in sizes)
    assert lib.fcn_convnet_logits_ld(ctypes.byref(c5)) == 128
    bad5 = _native.CnDesc(2, (ctypes.c_int32 * 5)(80, 40, 20, 10, 4), 10, 67, 1, 1e-5, 0.1, 0, 0, 5, 64)
    assert lib.fcn_convnet_sizes(ctypes.byref(bad5), ctypes.byref(sizes)) == 10001
    bad6 = _native.CnDesc(2, (ctypes.c_int32 * 5)(80, 40, 20, 10, 5), 10, 67, 1, 1e-5, 0.1, 0, 0, 6, 64)
    assert lib.fcn_convnet_sizes(ctypes.byref(bad6), ctypes.byref(sizes)) == 10001
    assert lib.fcn_convnet_pack(ctypes.byref(cd), None, None, None, None) == 10001
    assert lib.fcn_convnet_forward2(ctypes.byref(cd), None, None, (ctypes.c_void_p * 5)(), None, None, None, None) == 10001
    assert lib.fcn_adam_step_f32(None, None, None, None, 16, None, None, None) == 10001
    assert lib.fcn_adam_step_slots(0) == 0 and lib.fcn_adam_step_slots(3316780) == (3316780 // 4 + 511) // 512
    assert lib.fcn_stamp(None, None) == 10001
    idesc = _native.InpDesc(2, 64, 2, (ctypes.c_int32 * 4)(280, 140, 70, 35), (ctypes.c_double * 4)(0.25, 0.5, 1, 2), 70.0, 0, 0)
    assert lib.fcn_prepare_inputs(ctypes.byref(idesc), *([None] * 13), (ctypes.c_void_p * 4)(), *([None] * 7)) == 10001
    assert lib.fcn_det_loss_tail_rows(*([None] * 8), 2, 140, 12, 3, 1.0, 10.0, 20.0, 20.0, None, None, None) != 0


def test_sunrgbd_state_dict_matches_reference():
    """Keys, ORDER (it is the optimizer's parameter order in a checkpoint) and shapes of models/det_base_sunrgbd.py."""
    from frustum_convnet_amd.config import cfg, reset_cfg
    from frustum_convnet_amd import det_base_sunrgbd
    reset_cfg()
    g = load_golden("sunrgbd_b4_n1024")
    cfg.DATA.HEIGHT_HALF = tuple(float(x) for x in g["meta_strides"])
    cfg.DATA.DATASET_NAME = "SUNRGBD"
    m = det_base_sunrgbd.PointNetDet(3, num_vec=10, num_classes=2)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
    for k, s in zip(g["state_keys"], g["state_shapes"]):
        shape = tuple(int(x) for x in str(s).strip("()").split(",") if x.strip())
        assert tuple(sd[str(k)].shape) == shape, k
    reset_cfg()


def test_state_dict_keys_and_shapes_match_reference():
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    g = load_golden("car_b4_n512")
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
    assert len(sd) == 154
    for k, s in zip(g["state_keys"], g["state_shapes"]):
        assert str(tuple(sd[str(k)].shape)) == str(s), k
    assert sum(p.numel() for p in m.parameters()) == 3316777
    m.load_state_dict(golden_state_dict(g), strict=True)


def test_module_surface():
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    from frustum_convnet_amd.query_depth_point import QueryDepthPoint
    reset_cfg()
    q = QueryDepthPoint(0.25, 32)
    assert q.dis_z == 0.25 and q.nsample == 32 and len(list(q.parameters())) == 0
    pm = det_base.PointNetModule(0, [64, 64, 128], 0.5, 64, use_xyz=True, use_feature=True)
    assert pm.conv1[0].weight.shape == (64, 3, 1, 1) and pm.conv1[0].bias is None
    assert isinstance(pm.conv3[1], torch.nn.BatchNorm2d) and pm.use_feature is False
    f = det_base.PointNetFeat(3, 3)
    assert [n.nsample for n in (f.pointnet1, f.pointnet2, f.pointnet3, f.pointnet4)] == [32, 64, 64, 128]
    c = det_base.ConvFeatNet(128, 3)
    x = [torch.randn(2, 131, 280), torch.randn(2, 131, 140), torch.randn(2, 259, 70), torch.randn(2, 515, 35)]
    assert c(*x).shape == (2, 768, 140)
    with pytest.raises(NotImplementedError):
        det_base.PointNetModule(1, [64, 64, 128], 0.5, 64)


def test_hot_path_fails_loudly_on_cpu():
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    m = det_base.PointNetDet(3, num_vec=3)
    data = synth.to_torch(synth.make_batch(2, 64))
    with pytest.raises((RuntimeError, AssertionError)):
        m(data)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "frustum_convnet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
            assert "entry_ref" not in src, fn


def test_cfg_merge_semantics(tmp_path):
    from frustum_convnet_amd import config
    cfg = config.reset_cfg()
    config.merge_cfg_from_file(os.path.join(ROOT, "cfgs", "det_sample_people.yaml"))
    assert cfg.DATA.HEIGHT_HALF == (0.1, 0.2, 0.4, 0.8) and cfg.IOU_THRESH == 0.5 and cfg.DATA.PEOPLE_ONLY is True
    config.merge_cfg_from_list(["TRAIN.BATCH_SIZE", "8", "DATA.STRIDE", "(0.5, 1.0, 2.0, 4.0)", "OUTPUT_DIR", "out/x"])
    assert cfg.TRAIN.BATCH_SIZE == 8 and cfg.DATA.STRIDE == (0.5, 1.0, 2.0, 4.0) and cfg.OUTPUT_DIR == "out/x"
    p = tmp_path / "bad.yaml"
    p.write_text("DATA:\n  NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        config.merge_cfg_from_file(str(p))
    p.write_text("TRAIN:\n  BATCH_SIZE: 'abc'\n")
    with pytest.raises(ValueError):
        config.merge_cfg_from_file(str(p))
    p.write_text("TRAIN:\n  LR_STEPS: (20, 40)\n  MIN_LR: 1e-5\n")
    config.merge_cfg_from_file(str(p))
    assert cfg.TRAIN.LR_STEPS == [20, 40] and cfg.TRAIN.MIN_LR == 1e-5
    config.assert_and_infer_cfg()
    with pytest.raises(AttributeError):
        cfg.TRAIN.BATCH_SIZE = 4
    config.reset_cfg()


@pytest.mark.skipif(not os.path.isdir("/root/reference/cfgs"), reason="reference only exists in the build container")
@pytest.mark.parametrize("name", ["det_sample.yaml", "det_sample_people.yaml", "refine_car.yaml", "refine_people.yaml",
                                  "det_sample_sunrgbd.yaml"])
def test_reference_yaml_loads_unchanged(name):
    from frustum_convnet_amd import config
    cfg = config.reset_cfg()
    config.merge_cfg_from_file(os.path.join("/root/reference/cfgs", name))
    if "sunrgbd" in name:
        assert len(cfg.DATA.HEIGHT_HALF) == 5 and cfg.DATA.DATASET_NAME == "SUNRGBD" and cfg.IOU_THRESH == 0.25
        assert cfg.MODEL.FILE == "models/det_base_sunrgbd.py" and cfg.DATA.NUM_SAMPLES == 2048
    else:
        assert len(cfg.DATA.HEIGHT_HALF) == 4 and cfg.TRAIN.WEIGHT_DECAY == 0.0001
    config.reset_cfg()


def test_shipped_yamls_carry_the_reference_hot_path_keys():
    """Every cfg the reference ships for this path has a counterpart under cfgs/ with the same hot-path values (strides, window
    half heights, N, class selection, IoU threshold): cfgs/*.yaml is the surface north_star keeps."""
    from frustum_convnet_amd import config
    want = {"det_sample.yaml": ((0.25, 0.5, 1.0, 2.0), 1024, True, 0.7), "det_sample_people.yaml": ((0.1, 0.2, 0.4, 0.8), 1024, False, 0.5),
            "refine_car.yaml": ((0.1, 0.2, 0.4, 0.8), 512, True, 0.7), "refine_people.yaml": ((0.05, 0.1, 0.2, 0.4), 512, False, 0.5)}
    for name, (strides, n, car, thr) in want.items():
        cfg = config.reset_cfg()
        config.merge_cfg_from_file(os.path.join(ROOT, "cfgs", name))
        assert cfg.DATA.STRIDE == strides and cfg.DATA.HEIGHT_HALF == strides and cfg.DATA.NUM_SAMPLES == n, name
        assert cfg.DATA.CAR_ONLY is car and cfg.DATA.PEOPLE_ONLY is (not car) and cfg.IOU_THRESH == thr, name
        assert cfg.DATA.RTC is True and cfg.DATA.WITH_EXTRA_FEAT is False
    config.reset_cfg()


def test_own_sunrgbd_yaml_builds_the_five_scale_model():
    from frustum_convnet_amd import config, det_base_sunrgbd
    cfg = config.reset_cfg()
    config.merge_cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs",
                                            "det_sample_sunrgbd.yaml"))
    m = det_base_sunrgbd.PointNetDet(3, num_vec=10, num_classes=2)
    assert m.num_scales == 5 and m.reg_out.weight.shape == (67, 1024, 1) and m.num_size_cluster == 10
    assert [n.nsample for n in m.feat_net.nets] == [128, 128, 256, 256, 256]
    assert m.conv_net.block1_conv1[0].weight.shape == (64, 138, 3)
    assert m.conv_net.block5_deconv[0].weight.shape == (512, 256, 8)
    config.reset_cfg()


def test_loss_tail_matches_oracle_on_cpu():
    """The mask-weighted loss tail (no nonzero/indexing) equals the oracle's reference-style tail."""
    from oracle import det_ref
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g))
    m = det_base.PointNetDet(3, num_vec=3)
    m.fused_fcn = False            # explicit opt-in to the torch-op formulation of the tail (A/B path; the fused paths
    m.fused_loss = False           # refuse CPU tensors)
    cls_raw = torch.from_numpy(g["cls_train"]).requires_grad_(True)
    reg_raw = torch.from_numpy(g["reg_train"]).requires_grad_(True)
    # drive only the tail: replace the feature path by fixed logits
    m.feat_net.forward = lambda *a, **k: (None,) * 4
    m.conv_net.forward = lambda *a: torch.zeros(4, 768, 140)
    m.cls_out.forward = lambda x: cls_raw
    m.reg_out.forward = lambda x: reg_raw
    losses, metrics = m(data)
    ref = det_ref.loss_tail(cls_raw.detach(), reg_raw.detach(), data)
    for k, v in ref.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
    for nm, r in zip(g["loss_names"], g["loss_train"]):
        assert abs(float(losses[str(nm)]) - r) <= 1e-4 * max(1.0, abs(r)), nm
    # gradient of the tail w.r.t. the logits agrees too
    losses["total_loss"].backward()
    c2 = cls_raw.detach().clone().requires_grad_(True)
    r2 = reg_raw.detach().clone().requires_grad_(True)
    det_ref.loss_tail(c2, r2, data)["total_loss"].backward()
    assert torch.allclose(cls_raw.grad, c2.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(reg_raw.grad, r2.grad, rtol=1e-4, atol=1e-6)
    assert 0.0 <= float(metrics["cls_acc"]) <= 1.0


def test_flat_train_state_layout_on_cpu():
    """FlatTrainState re-homes parameters/gradients into flat buffers (heads adjacent, 16-byte aligned starts); the
    optimiser step itself is a HIP kernel and refuses to run on the CPU."""
    import pytest
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    from frustum_convnet_amd.train_state import FlatTrainState
    from frustum_convnet_amd.fcn_fused import _adjacent
    reset_cfg()
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st = FlatTrainState(m, lr=1e-3, weight_decay=1e-4, world=4)
    assert abs(float(st.hyper[5]) - 0.25) < 1e-7
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    named = dict(m.named_parameters())
    assert len(st.params) == len(named) == 79 and sum(p.numel() for p in st.params) == 3316777
    for p, o in zip(st.params, st.offsets):
        assert p.data_ptr() == st.flat.data_ptr() + 4 * o and p.grad.data_ptr() == st.grad.data_ptr() + 4 * o
    Wh = _adjacent(named["cls_out.weight"], named["reg_out.weight"])
    assert Wh is not None and Wh.shape[0] == 41 and Wh.data_ptr() == named["cls_out.weight"].data_ptr()
    assert torch.equal(Wh[2:], named["reg_out.weight"])
    bh = _adjacent(named["cls_out.bias"], named["reg_out.bias"])
    assert bh is not None and bh.shape == (41,)
    assert _adjacent(named["reg_out.weight"], named["cls_out.weight"]) is None
    with pytest.raises(RuntimeError):
        st.adam_step()
    st.release()
    assert all(p.grad is None for p in st.params)


def test_optimizer_state_interchanges_with_torch_adam():
    """the reference checkpoints optim.Adam(model.parameters()).state_dict() (train/train_net_det.py:353,387), whose
    per-parameter state is indexed in model.parameters() order (..., reg_out.weight, reg_out.bias, cls_out.weight,
    cls_out.bias).  FlatTrainState keeps the head tensors adjacent in ITS buffer but emits / accepts that order."""
    import pytest
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    from frustum_convnet_amd.train_state import FlatTrainState
    reset_cfg()
    torch.manual_seed(3)
    ref = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    opt = torch.optim.Adam(ref.parameters(), lr=2e-3, weight_decay=1e-4)
    for p in ref.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    for p in ref.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    sd_ref = opt.state_dict()
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    st = FlatTrainState(m, lr=1e-3)
    st.load_state_dict(sd_ref)                          # a reference optimizer checkpoint: no 'names', model order
    views = {n: (st.exp_avg[o:o + p.numel()].view(p.shape), st.exp_avg_sq[o:o + p.numel()].view(p.shape))
             for n, p, o in zip(st.names, st.params, st.offsets)}
    for n, p in ref.named_parameters():
        assert torch.equal(views[n][0], opt.state[p]["exp_avg"]), n
        assert torch.equal(views[n][1], opt.state[p]["exp_avg_sq"]), n
    assert int(st._step_slots.min()) == 2 and abs(float(st.hyper[0]) - 2e-3) < 1e-9
    # and back: torch.optim.Adam(model.parameters()) of the reference loads what this object writes
    sd = st.state_dict()
    assert sd["names"] == [n for n, _ in ref.named_parameters()]
    opt2 = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt2.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    for p in ref.parameters():
        assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"])
        assert float(opt2.state[p]["step"]) == 2.0
    # by-name matching survives a permuted dict; a wrong shape is refused before anything is copied
    perm = list(range(len(sd["names"])))[::-1]
    sd_perm = {"state": {k: sd["state"][i] for k, i in enumerate(perm)}, "param_groups": sd["param_groups"],
               "names": [sd["names"][i] for i in perm]}
    st2 = FlatTrainState(det_base.PointNetDet(3, num_vec=3, num_classes=2), lr=1e-3)
    st2.load_state_dict(sd_perm)
    assert torch.equal(st2.exp_avg, st.exp_avg) and torch.equal(st2.exp_avg_sq, st.exp_avg_sq)
    bad = {"state": dict(sd["state"]), "param_groups": sd["param_groups"]}
    bad["state"][0], bad["state"][3] = bad["state"][3], bad["state"][0]
    with pytest.raises(ValueError, match="shape"):
        st2.load_state_dict(bad)
    # a LATER entry without its 'step' (or with another step count) is refused BEFORE the first copy -- the moments
    # of the earlier entries stay what they were
    last = max(sd["state"].keys())
    for broken in ({k: v for k, v in sd["state"][last].items() if k != "step"}, dict(sd["state"][last], step=torch.tensor(5.0))):
        bad2 = {"state": dict(sd["state"]), "param_groups": sd["param_groups"], "names": sd["names"]}
        bad2["state"][last] = broken
        st3 = FlatTrainState(det_base.PointNetDet(3, num_vec=3, num_classes=2), lr=1e-3)
        st3.exp_avg.fill_(0.25)
        with pytest.raises(ValueError, match="step"):
            st3.load_state_dict(bad2)
        assert float(st3.exp_avg.min()) == 0.25 and float(st3.exp_avg.max()) == 0.25


def test_refine_builder_refuses_the_non_rtc_geometry():
    """the refine loader's kernel implements cfg.DATA.RTC = True (every shipped refine cfg); with RTC False the
    reference builds its windows on the un-rotated predicted box (provider_sample_refine.py:225-262) -- refuse, do not
    silently produce the other geometry."""
    import pytest
    from frustum_convnet_amd import config, inputs
    cfg = config.reset_cfg()
    cfg.DATA.RTC = False
    with pytest.raises(NotImplementedError, match="RTC"):
        inputs.RefineInputBuilder(512, strides=(0.1, 0.2, 0.4, 0.8))
    config.reset_cfg()
    inputs.RefineInputBuilder(512, strides=(0.1, 0.2, 0.4, 0.8))      # the default (RTC True) constructs



def test_bench_gpus_n_never_prints_a_line_for_fewer_ranks(tmp_path):
    """`python bench.py --gpus 8` without a launcher used to measure ONE GPU and print n_gpus 1.  Now --gpus N with
    WORLD_SIZE unset launches N ranks itself and refuses -- non-zero exit, no JSON line -- on a box with fewer than N GPUs (here:
    none); with a launcher's WORLD_SIZE that disagrees with --gpus it refuses as well, for N = 1 too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FCN_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout and "refusing" in r.stderr
    for gpus, world in (("1", "2"), ("2", "1"), ("4", "2")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                           env=dict(env, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and '"metric"' not in r.stdout, (gpus, world, r.stdout[-300:], r.stderr[-300:])


def test_pmc_traffic_is_keyed_by_configuration(tmp_path, monkeypatch):
    """every non-car bench line printed the CAR configuration's HBM counters.  The record is now keyed by
    configuration and tied to the kernel sources: another configuration, or other sources, give None + the reason (-> `traffic: null`)."""
    import json
    import bench
    os.makedirs(tmp_path / "profiles")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "abc")
    rec = {"source_hash": "abc", "configs": {"car": {"step": {"bytes_per_step": 10}, "entries": {"fcn_pn_forward": {"bytes_per_launch": 3}}}}}
    json.dump(rec, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    assert bench.pmc_traffic("step", cfg_name="car") == ({"bytes_per_step": 10}, None)
    assert bench.pmc_traffic("entry", "fcn_pn_forward", "car")[0] == {"bytes_per_launch": 3}
    got, why = bench.pmc_traffic("step", cfg_name="refine")
    assert got is None and "refine" in why and "car" in why
    got, why = bench.pmc_traffic("entry", "fcn_convnet_backward", "car")
    assert got is None and "fcn_convnet_backward" in why
    got, why = bench.pmc_traffic("step", cfg_name="car", prec="bf16")
    assert got is None and "operand mode" in why
    monkeypatch.setattr(bench, "source_hash", lambda: "other")
    got, why = bench.pmc_traffic("step", cfg_name="car")
    assert got is None and "other kernel sources" in why
