"""-m gpu: on-device batch construction (C-ABI fcn_prepare_inputs through frustum_convnet_amd.inputs.InputBuilder) against
the golden vectors of the reference's own ProviderDataset and against the oracle on seeded cases."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FLOAT_KEYS = ("point_cloud", "center_ref1", "center_ref2", "center_ref3", "center_ref4", "box3d_center",
              "box3d_heading", "box3d_size", "rot_angle")


def _golden():
    return np.load(os.path.join(HERE, "golden", "inputs_kitti_b6.npz"))


def _builder(g, **kw):
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.inputs import InputBuilder
    reset_cfg()
    return InputBuilder(int(g["meta_npoint"]), tuple(g["meta_strides"]), float(g["meta_max_depth"]), **kw)


def _check(out, want, prefix=""):
    assert torch.equal(out["cls_label"].cpu(), torch.from_numpy(np.asarray(want[prefix + "cls_label"])))
    assert torch.equal(out["seg_label"].cpu(), torch.from_numpy(np.asarray(want[prefix + "seg_label"])))
    for k in FLOAT_KEYS:
        ref = np.asarray(want[prefix + k])
        got = out[k].cpu().numpy().reshape(ref.shape)
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
        # fp64 arithmetic rounded to fp32 on both sides (device libm cos/sin vs numpy's may differ in the last fp64 bit)
        assert d <= 1e-6 * max(1.0, np.abs(ref).max()), (k, d)


def test_golden_batch_with_recorded_draws():
    from frustum_convnet_amd.inputs import records_from_fixture
    g = _golden()
    b = _builder(g, random_flip=True, random_shift=True)
    out = b.build(records_from_fixture(g), draws=(g["draw_choice"], g["draw_coin"], g["draw_normal"]))
    _check(out, g, "ref_")
    assert torch.equal(out["size_class"].cpu(), torch.from_numpy(g["ref_size_class"]))
    assert torch.equal(out["one_hot"].cpu(), torch.from_numpy(g["ref_one_hot"]))
    for k in FLOAT_KEYS:
        assert out[k].dtype == torch.float32 and tuple(out[k].shape) == g["ref_" + k].shape, k


def test_same_numpy_seed_reproduces_the_reference_batch():
    """Draws taken like the reference takes them (choice, coin, randn per sample) from the same seed: same batch."""
    from frustum_convnet_amd.inputs import records_from_fixture
    g = _golden()
    b = _builder(g, random_flip=True, random_shift=True)
    np.random.seed(4242)                        # tests/golden/make_golden_inputs.py seeds the reference run with this
    out = b.build(records_from_fixture(g))
    _check(out, g, "ref_")


@pytest.mark.parametrize("flip,shift", [(False, False), (True, False), (False, True)])
def test_against_oracle_without_augmentation_and_nearest_fallback(flip, shift):
    """Seeded records with tiny boxes (no centre inside the half box -> nearest-centre fallback) and switched-off
    augmentations, against oracle/inputs_ref.py."""
    from oracle import inputs_ref
    from frustum_convnet_amd.inputs import records_from_fixture, draw
    g = {k: np.array(v) for k, v in _golden().items()}
    g["size"] = g["size"] * np.array([0.02, 0.02, 0.02])          # boxes far smaller than the centre spacing
    g["box3d_corners"] = g["box3d_corners"].mean(1, keepdims=True) + 0.02 * (g["box3d_corners"] - g["box3d_corners"].mean(1, keepdims=True))
    rng = np.random.RandomState(11)
    choice, coin, normal = draw(g["raw_counts"], int(g["meta_npoint"]), True, True, rng)
    g["draw_choice"], g["draw_coin"], g["draw_normal"] = choice, coin, normal
    want = inputs_ref.prepare_batch(g, tuple(g["meta_strides"]), float(g["meta_max_depth"]), flip, shift)
    assert ((want["cls_label"] == 1).sum(1) == 1).all() and (want["cls_label"] == -1).sum() == 0
    b = _builder(g, random_flip=flip, random_shift=shift)
    out = b.build(records_from_fixture(g), draws=(choice, coin, normal))
    _check(out, want)


def test_built_batch_feeds_the_model():
    """The dict goes straight into PointNetDet.forward (train step on device-built inputs)."""
    from frustum_convnet_amd import det_base, synth
    from frustum_convnet_amd.inputs import records_from_fixture
    g = _golden()
    b = _builder(g, random_flip=True, random_shift=True)
    data = b.build(records_from_fixture(g), draws=(g["draw_choice"], g["draw_coin"], g["draw_normal"]))
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    synth.fill_state_dict(m.state_dict(), seed=7)
    m = m.cuda().train()
    losses, _ = m(data)
    losses["total_loss"].backward()
    assert torch.isfinite(losses["total_loss"]) and float(losses["total_loss"]) > 0


def test_synthetic_records_against_oracle_and_launch_into_given_buffers():
    """synth.make_records (bench.py's raw frustum records) through upload() + launch(out=...) -- the capturable, allocation-free form
    -- against oracle/inputs_ref.py on the same records and draws; a second launch into the same buffers reproduces them bit for bit,
    and build() (= upload + launch) equals the split form."""
    from oracle import inputs_ref
    from frustum_convnet_amd import synth
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.inputs import InputBuilder, draw
    reset_cfg()
    B, N, strides = 5, 512, (0.25, 0.5, 1.0, 2.0)
    recs = synth.make_records(B, seed=77)
    counts = [len(r["points"]) for r in recs]
    draws = draw(counts, N, True, True, np.random.RandomState(5))
    rec = {"raw_points": np.concatenate([r["points"] for r in recs]), "raw_seg": np.concatenate([r["seg"] for r in recs]),
           "raw_counts": np.asarray(counts), "box2d": np.stack([r["box2d"] for r in recs]), "P": np.stack([r["P"] for r in recs]),
           "box3d_corners": np.stack([r["box3d"] for r in recs]), "heading": np.asarray([r["heading"] for r in recs]),
           "size": np.stack([r["size"] for r in recs]), "frustum_angle": np.asarray([r["frustum_angle"] for r in recs]),
           "draw_choice": draws[0], "draw_coin": draws[1], "draw_normal": draws[2]}
    want = inputs_ref.prepare_batch(rec, strides, 70.0)
    assert ((want["cls_label"] == 1).sum(1) >= 1).all()
    b = InputBuilder(N, strides, 70.0, random_flip=True, random_shift=True)
    t = b.upload(recs, draws, with_seg=True)
    out = b.alloc(B, with_seg=True)
    ptrs = {k: v.data_ptr() for k, v in out.items()}
    written = list(out)                                   # (launch() adds the uploaded size_class / one_hot tensors to the dict)
    got = b.launch(t, out=out)
    assert all(got[k].data_ptr() == p for k, p in ptrs.items())
    _check(got, want)
    first = {k: v.clone() for k, v in got.items()}
    for k in written:
        out[k].zero_()
    got = b.launch(t, out=out)
    for k, v in first.items():
        assert torch.equal(got[k], v), k
    whole = b.build(recs, draws)
    for k, v in first.items():
        assert torch.equal(whole[k], v), k
    assert b.algorithmic_bytes(B, with_seg=True, pt_stride=4) > 0


def test_batch_built_on_the_prefetch_branch_feeds_the_same_step():
    """PointNetDet.prefetch(data, before=launch): the NEXT batch is produced on the prefetch branch, in front of its grouping front, into
    the tensors the next forward reads (bench.py's build_inputs line).  The losses of that forward equal the losses on a batch built
    ahead of time, bit for bit; a point cloud that would have to be copied first is refused."""
    from frustum_convnet_amd import det_base, synth
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.inputs import InputBuilder, draw
    reset_cfg()
    B, N, strides = 4, 512, (0.25, 0.5, 1.0, 2.0)
    recs = synth.make_records(B, seed=9)
    draws = draw([len(r["points"]) for r in recs], N, True, True, np.random.RandomState(2))
    b = InputBuilder(N, strides, 70.0, random_flip=True, random_shift=True)
    t = b.upload(recs, draws, with_seg=False)

    def model():
        m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
        synth.fill_state_dict(m.state_dict(), seed=7)
        return m.cuda().train()

    ready = b.launch(t)                                   # built ahead of time
    m0 = model()
    l0, _ = m0(ready)
    want = {k: float(v) for k, v in l0.items()}
    m1 = model()
    cur = b.launch(t)                                     # this step's batch
    nxt = b.alloc(B, with_seg=False)                      # the next one: NOT built yet
    for v in nxt.values():
        v.zero_()
    nxt["size_class"], nxt["one_hot"] = t["size_class"], t["one_hot"]        # (what launch() will hand back as well)
    m1.next_batch = nxt
    m1.next_batch_build = lambda: b.launch(t, out=nxt)
    l1, _ = m1(cur)
    m1.backward(l1["total_loss"])                         # (joins the prefetch branch)
    assert m1.feat_net._prefetched is not None
    torch.cuda.synchronize()
    assert float(nxt["point_cloud"].abs().max()) > 0      # the branch built it
    # a second model with the same weights consumes batch `nxt` without any prefetch: same losses as on `ready`
    l2, _ = m0.__class__.forward(model(), nxt)
    assert {k: float(v) for k, v in l2.items()} == want
    # and m1 itself consumes its prefetched front (running statistics moved by its first step, so only finiteness is checked here)
    l3, _ = m1(nxt)
    assert m1.feat_net._prefetched is None and all(torch.isfinite(v) for v in l3.values())
    with pytest.raises(RuntimeError, match="prefetch branch"):
        four = torch.cat([nxt["point_cloud"], nxt["point_cloud"][:, :1]], 1).contiguous()
        m1.prefetch(dict(nxt, point_cloud=four), before=lambda: None)


def test_cpu_builder_fails_loudly():
    g = _golden()
    from frustum_convnet_amd.inputs import records_from_fixture
    b = _builder(g, device="cpu")
    with pytest.raises(RuntimeError):
        b.build(records_from_fixture(g))


@pytest.mark.parametrize("B,N,stride3", [(1, 100, True), (3, 257, False), (6, 1024, True)])
def test_ragged_and_odd_sizes_against_oracle(B, N, stride3):
    """Single frustum, N not a multiple of the workgroup size, xyz-only records (pt_stride 3), N larger than some records
    (replacement) -- against oracle/inputs_ref.py; also the no-segmentation call."""
    from oracle import inputs_ref
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.inputs import InputBuilder, records_from_fixture, draw
    g = {k: np.array(v) for k, v in _golden().items()}
    recs = records_from_fixture(g)[:B]
    if stride3:
        for r in recs:
            r["points"] = np.ascontiguousarray(r["points"][:, :3])
    rng = np.random.RandomState(100 + N)
    counts = [len(r["points"]) for r in recs]
    choice, coin, normal = draw(counts, N, True, True, rng)
    sub = {k: v for k, v in g.items()}
    offs = np.concatenate([[0], np.cumsum(g["raw_counts"])])
    sub["raw_counts"] = g["raw_counts"][:B]
    sub["raw_points"] = g["raw_points"][:offs[B], :3] if stride3 else g["raw_points"][:offs[B]]
    sub["raw_seg"] = g["raw_seg"][:offs[B]]
    for k in ("box2d", "P", "box3d_corners", "heading", "size", "frustum_angle"):
        sub[k] = g[k][:B]
    sub["draw_choice"], sub["draw_coin"], sub["draw_normal"] = choice, coin, normal
    want = inputs_ref.prepare_batch(sub, tuple(g["meta_strides"]), float(g["meta_max_depth"]))
    reset_cfg()
    b = InputBuilder(N, tuple(g["meta_strides"]), float(g["meta_max_depth"]), random_flip=True, random_shift=True)
    out = b.build(recs, draws=(choice, coin, normal))
    _check(out, want)
    assert out["point_cloud"].shape == (B, 3, N)
    out2 = b.build(recs, draws=(choice, coin, normal), with_seg=False)
    assert "seg_label" not in out2 and torch.equal(out2["point_cloud"], out["point_cloud"])
    assert torch.equal(out2["cls_label"], out["cls_label"])


def test_refine_builder_matches_reference_batch():
    """fcn_prepare_inputs_refine vs the reference's refine ProviderDataset + collate_fn outputs (golden fixture): labels and
    per-sample window counts exact, floats <= 1e-6, edge padding included."""
    import numpy as np
    import torch
    from helpers import load_golden
    from frustum_convnet_amd import inputs
    from frustum_convnet_amd.config import reset_cfg
    reset_cfg()
    g = load_golden("inputs_refine_b6")
    b = inputs.RefineInputBuilder(int(g["meta_npoint"]), strides=tuple(g["meta_strides"]), random_flip=True, random_shift=True)
    out = b.build(inputs.refine_records_from_fixture(g), draws=(g["draw_choice"], g["draw_coin"], g["draw_normal"]))
    torch.cuda.synchronize()
    assert np.array_equal(out["lens"].cpu().numpy(), g["ref_lens"])
    for k in ("point_cloud", "cls_label", "box3d_center", "box3d_heading", "box3d_size", "size_class", "center_ref1",
              "center_ref2", "center_ref3", "center_ref4", "rot_angle", "ref_center", "one_hot"):
        got, ref = out[k].cpu().numpy(), g["ref_" + k]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if got.dtype.kind == "f":
            assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= 1e-6, k
        else:
            assert np.array_equal(got, ref), k
    # inference records (no labels): same points / centres without flip and shift
    out2 = b.build(inputs.refine_records_from_fixture(g), draws=(g["draw_choice"], g["draw_coin"] * 0, g["draw_normal"] * 0),
                   with_labels=False)
    assert "cls_label" not in out2 and out2["center_ref1"].shape == out["center_ref1"].shape
    #  from_rgb_detection records carry the 2-D detector's score: it travels as 'rgb_prob' (B,1), as
    # provider_sample_refine.py:228-238 returns it; records without one default to 1 (test_net_det.py:203-205)
    recs = inputs.refine_records_from_fixture(g)
    for i, r in enumerate(recs):
        r["prob"] = 0.25 + 0.1 * i
    out3 = b.build(recs, draws=(g["draw_choice"], g["draw_coin"] * 0, g["draw_normal"] * 0), with_labels=False)
    assert np.allclose(out3["rgb_prob"].cpu().numpy().ravel(), [0.25 + 0.1 * i for i in range(len(recs))])
    assert np.allclose(out2["rgb_prob"].cpu().numpy(), 1.0) and out2["rgb_prob"].shape == (len(recs), 1)
    assert "rgb_prob" not in out
    # the built batch feeds the model (variable L incl. L4 = 3)
    from frustum_convnet_amd import det_base
    from frustum_convnet_amd.config import cfg
    cfg.DATA.HEIGHT_HALF = tuple(float(x) for x in g["meta_strides"]); cfg.DATA.STRIDE = cfg.DATA.HEIGHT_HALF
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2).cuda().train()
    feed = {k: v for k, v in out.items() if k not in ("lens", "rot_angle", "ref_center")}
    losses, _ = m(feed)
    assert torch.isfinite(losses["total_loss"])


def test_sunrgbd_builder_matches_reference_batch():
    """fcn_prepare_inputs_sunrgbd (five strides, centres through K / Rtilt, depth + height shift) against the outputs of the
    reference's own provider_sample_sunrgbd.ProviderDataset, with the recorded draws and with the same numpy seed."""
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.inputs import SunrgbdInputBuilder, sunrgbd_records_from_fixture
    reset_cfg()
    g = np.load(os.path.join(HERE, "golden", "inputs_sunrgbd_b6.npz"))
    b = SunrgbdInputBuilder(int(g["meta_npoint"]), tuple(g["meta_strides"]), float(g["meta_max_depth"]),
                            random_flip=True, random_shift=True)
    recs = sunrgbd_records_from_fixture(g)
    keys = FLOAT_KEYS + ("center_ref5",)
    for draws in ((g["draw_choice"], g["draw_coin"], g["draw_normal"], g["draw_hshift"]), None):
        if draws is None:
            np.random.seed(777)                 # make_golden_inputs_sunrgbd.py seeds the reference run with this
        out = b.build(recs, draws=draws)
        assert torch.equal(out["cls_label"].cpu(), torch.from_numpy(g["ref_cls_label"]))
        assert torch.equal(out["seg_label"].cpu(), torch.from_numpy(g["ref_seg_label"]))
        assert torch.equal(out["size_class"].cpu(), torch.from_numpy(g["ref_size_class"]))
        assert torch.equal(out["one_hot"].cpu(), torch.from_numpy(g["ref_one_hot"]))
        for k in keys:
            ref = g["ref_" + k]
            got = out[k].cpu().numpy()
            assert got.shape == ref.shape and out[k].dtype == torch.float32, k
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
            assert d <= 1e-6 * max(1.0, np.abs(ref).max()), (k, d)


def test_sunrgbd_batch_feeds_the_five_scale_model():
    from frustum_convnet_amd.config import cfg, reset_cfg
    from frustum_convnet_amd import det_base_sunrgbd, synth
    from frustum_convnet_amd.inputs import SunrgbdInputBuilder, sunrgbd_records_from_fixture
    reset_cfg()
    g = np.load(os.path.join(HERE, "golden", "inputs_sunrgbd_b6.npz"))
    cfg.DATA.HEIGHT_HALF = tuple(float(x) for x in g["meta_strides"])
    cfg.DATA.STRIDE = cfg.DATA.HEIGHT_HALF
    cfg.DATA.DATASET_NAME = "SUNRGBD"
    cfg.DATA.MAX_DEPTH = float(g["meta_max_depth"])
    b = SunrgbdInputBuilder(int(g["meta_npoint"]), random_flip=True, random_shift=True)
    data = b.build(sunrgbd_records_from_fixture(g), draws=(g["draw_choice"], g["draw_coin"], g["draw_normal"], g["draw_hshift"]))
    m = det_base_sunrgbd.PointNetDet(3, num_vec=10, num_classes=2)
    synth.fill_state_dict(m.state_dict(), seed=7)
    m = m.cuda().train()
    losses, metrics = m(data)
    losses["total_loss"].backward()
    assert all(torch.isfinite(v).all() for v in losses.values())
    assert float(m.last_num_fg) == float((data["cls_label"] == 1).sum())
    reset_cfg()
