"""The input-construction oracle (oracle/inputs_ref.py) against the golden vectors captured from the reference's own
ProviderDataset (tests/golden/make_golden_inputs.py), plus the branches the fixture does not reach."""
import os

import numpy as np

from oracle import inputs_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return np.load(os.path.join(HERE, "golden", "inputs_kitti_b6.npz"))


def test_oracle_matches_reference_provider():
    g = load()
    out = inputs_ref.prepare_batch(g, tuple(g["meta_strides"]), float(g["meta_max_depth"]))
    assert out["point_cloud"].shape == g["ref_point_cloud"].shape
    # integer outputs: exact
    assert np.array_equal(out["cls_label"], g["ref_cls_label"])
    assert np.array_equal(out["seg_label"], g["ref_seg_label"])
    # float outputs: fp64 arithmetic rounded to fp32 on both sides; BLAS may fuse the 2-term rotation dot product
    for k in ("point_cloud", "center_ref1", "center_ref2", "center_ref3", "center_ref4", "box3d_center",
              "box3d_heading", "box3d_size", "rot_angle"):
        ref = g["ref_" + k]
        got = out[k].reshape(ref.shape)
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
        assert d <= 1e-6 * max(1.0, np.abs(ref).max()), (k, d)
    # the fixture exercises both resampling modes and both flip outcomes
    assert (g["raw_counts"] < g["meta_npoint"]).any() and (g["raw_counts"] > g["meta_npoint"]).any()
    assert (g["draw_coin"] > 0.5).any() and (g["draw_coin"] <= 0.5).any()


def test_nearest_centre_fallback_and_label_precedence():
    """No centre inside the half box -> the nearest one becomes the positive (provider_sample.py:284-287)."""
    ref = np.stack([np.zeros(10), np.zeros(10), np.arange(10) * 2.0 + 1.0], 1)
    center = np.array([0.3, 0.0, 8.0])
    lab = inputs_ref.generate_labels(center, np.array([1.0, 1.0, 1.0]), 0.0, ref)      # half box 0.5 wide: nobody inside
    assert lab.sum() == 1 and lab[np.argmin(np.abs(ref[:, 2] - 8.0))] == 1 and (lab == -1).sum() == 0
    lab = inputs_ref.generate_labels(np.array([0.0, 0.0, 8.0]), np.array([9.0, 9.0, 4.0]), np.pi / 2, ref)
    # heading pi/2: the box's length (9) lies along z -> centres 3.5..12.5 inside the full box, 5.75..10.25 in the half
    assert lab.tolist() == [0, 0, -1, 1, 1, -1, 0, 0, 0, 0]


def test_in_box_matches_delaunay_hull_on_random_points():
    """The closed-form test equals the reference's scipy Delaunay hull membership away from the faces."""
    from scipy.spatial import Delaunay
    rng = np.random.RandomState(5)
    for _ in range(5):
        center = rng.uniform(-2, 2, 3) + np.array([0, 0, 20.0])
        dims = rng.uniform(1.0, 4.0, 3)
        ang = rng.uniform(-np.pi, np.pi)
        l, w, h = dims
        c, s = np.cos(ang), np.sin(ang)
        xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
        yc = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
        zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
        corners = np.stack([c * xc + s * zc + center[0], yc + center[1], -s * xc + c * zc + center[2]], 1)
        p = center[None, :] + rng.uniform(-3, 3, (400, 3))
        want = Delaunay(corners).find_simplex(p) >= 0
        assert np.array_equal(inputs_ref.in_box(p, center, dims, ang), want)


def test_host_draws_follow_the_reference_rng_order():
    """inputs.draw() consumes numpy's global RNG in the reference's order (choice, coin, randn per sample), so the same
    seed gives the draws the reference's ProviderDataset took when the fixture was made (seed 4242)."""
    from frustum_convnet_amd.inputs import draw
    g = load()
    np.random.seed(4242)
    choice, coin, normal = draw(g["raw_counts"], int(g["meta_npoint"]), True, True)
    assert np.array_equal(choice, g["draw_choice"])
    assert np.array_equal(coin, g["draw_coin"]) and np.array_equal(normal, g["draw_normal"])
    # a record with fewer points than NUM_SAMPLES is resampled WITH replacement, the others without
    for n, c in zip(g["raw_counts"], choice):
        assert (len(np.unique(c)) == len(c)) == (n >= len(c))


def test_refine_oracle_matches_reference_dataset_and_collate():
    """oracle/inputs_ref.py refine restatement vs the reference's own refine ProviderDataset + collate_fn outputs
    (tests/golden/make_golden_inputs_refine.py): per-sample L differs, the padded batch must match exactly."""
    import numpy as np
    from oracle import inputs_ref
    from helpers import load_golden
    g = load_golden("inputs_refine_b6")
    out = inputs_ref.prepare_batch_refine(g, tuple(g["meta_strides"]))
    assert len(set(int(v) for v in g["ref_lens"][:, 0])) > 1              # the fixture does exercise the padding
    for k, v in out.items():
        ref = g["ref_" + k]
        assert v.shape == ref.shape, k
        if v.dtype.kind == "f":
            assert np.abs(v.astype(np.float64) - ref.astype(np.float64)).max() <= 1e-6, k
        else:
            assert np.array_equal(v, ref), k


def test_sunrgbd_oracle_matches_reference_provider():
    """oracle/inputs_ref.py SUN-RGBD restatement vs the reference's own provider_sample_sunrgbd.ProviderDataset outputs
    (tests/golden/make_golden_inputs_sunrgbd.py): five strides through K / Rtilt, depth + height shift."""
    g = np.load(os.path.join(HERE, "golden", "inputs_sunrgbd_b6.npz"))
    out = inputs_ref.prepare_batch_sunrgbd(g, tuple(g["meta_strides"]), float(g["meta_max_depth"]))
    assert np.array_equal(out["cls_label"], g["ref_cls_label"])
    assert np.array_equal(out["seg_label"], g["ref_seg_label"])
    for k in ("point_cloud", "center_ref1", "center_ref2", "center_ref3", "center_ref4", "center_ref5", "box3d_center",
              "box3d_heading", "box3d_size", "rot_angle"):
        ref = g["ref_" + k]
        got = out[k].reshape(ref.shape)
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
        assert d <= 1e-6 * max(1.0, np.abs(ref).max()), (k, d)
    assert (g["raw_counts"] < g["meta_npoint"]).any() and (g["raw_counts"] > g["meta_npoint"]).any()
    assert (g["draw_coin"] > 0.5).any() and (g["draw_coin"] <= 0.5).any()
    # the draws follow the loader's order: same seed, same draws (and replacement only for the short frustum)
    from frustum_convnet_amd.inputs import draw_sunrgbd
    np.random.seed(777)
    choice, coin, normal, hshift = draw_sunrgbd(g["raw_counts"], int(g["meta_npoint"]), True, True)
    assert np.array_equal(choice, g["draw_choice"]) and np.array_equal(coin, g["draw_coin"])
    assert np.array_equal(normal, g["draw_normal"]) and np.array_equal(hshift, g["draw_hshift"])
