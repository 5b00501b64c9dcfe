"""Pins the rotated-box oracle (oracle/box_ref.py) against fixtures generated from the reference's own python
(tests/golden/make_golden_iou.py: utils/box_util.py box3d_iou_pair, ops/pybind11/rbbox_iou.py cube_nms_np,
datasets/provider_sample.py from_prediction_to_label_format), and checks the float arithmetic of the device clip core
(csrc/box_iou.h) by compiling that header with g++ (tests/host_harness) -- no GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import box_ref
from helpers import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))


def closed_form(a, b, kind):
    """coincident footprints: kind 4 identical -> (1, 1); kind 5 y-shifted copy -> (1, ov / (2h - ov))."""
    if kind == 4:
        return 1.0, 1.0
    h = a[5]
    ov = max(0.0, h - abs(a[1] - b[1]))
    return 1.0, ov / (2 * h - ov)


def test_iou_pair_matches_reference_python():
    g = load_golden("box_iou_pairs")
    a, b, ref, kind = g["boxes_a"], g["boxes_b"], g["ious"], g["kind"]
    got = box_ref.iou_pair(box_ref.boxes3d2corners(a), box_ref.boxes3d2corners(b))
    ok = np.isfinite(ref[:, 0])
    assert ok.sum() >= 250
    assert np.abs(got[ok] - ref[ok]).max() < 1e-9
    for i in np.nonzero(~ok)[0]:
        e2, e3 = closed_form(a[i], b[i], int(kind[i]))
        assert abs(got[i, 0] - e2) < 1e-9 and abs(got[i, 1] - e3) < 1e-9, (i, got[i], e2, e3)
    assert (got >= 0).all() and (got <= 1 + 1e-12).all()


def test_cube_nms_matches_reference_loop():
    g = load_golden("box_nms_cases")
    for c in range(int(g["ncase"])):
        keep = box_ref.cube_nms(g["dets%d" % c], float(g["thr%d" % c]))
        assert keep == [int(v) for v in g["keep%d" % c]], c


def test_label_format_matches_reference():
    g = load_golden("box_nms_cases")
    m = len(g["lf_rot"])
    for i in range(m):
        probs = np.array([[0.2, 0.8]])
        rows, idx = box_ref.decode_detections(probs, g["lf_center"][i:i + 1], g["lf_angle"][i:i + 1], g["lf_size"][i:i + 1],
                                              g["lf_rot"][i], g["lf_ref"][i], 0.0)
        h, w, l, tx, ty, tz, ry = g["lf_out"][i]
        assert np.allclose(rows[0], [tx, ty, tz, l, w, h, ry, 0.8], rtol=0, atol=1e-12)


def test_decode_fallback_and_filter():
    probs = np.array([[0.9, 0.1], [0.6, 0.4], [0.7, 0.3]])
    ctr = np.zeros((3, 3)); ang = np.zeros(3); size = np.ones((3, 3))
    rows, idx = box_ref.decode_detections(probs, ctr, ang, size, 0.0, np.zeros(3), 0.5)
    assert idx == [1] and abs(rows[0, 7] - 0.9) < 1e-12          # no foreground position: the arg-max of p_fg is taken
    size[1, 2] = 0.001
    rows, idx = box_ref.decode_detections(probs, ctr, ang, size, 0.0, np.zeros(3), 0.5)
    assert idx == [] and rows.shape == (0, 8)                    # too-small boxes are dropped


@pytest.fixture(scope="module")
def host_lib():
    out = os.path.join(HERE, "host_harness", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libbox_iou_host.so")
    src = os.path.join(HERE, "host_harness", "box_iou_host.cpp")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def test_device_clip_core_arithmetic_on_host(host_lib):
    """csrc/box_iou.h in float32 vs the float64 oracle (and through it the reference's python) on every fixture pair."""
    g = load_golden("box_iou_pairs")
    a, b, kind = g["boxes_a"].astype(np.float32), g["boxes_b"].astype(np.float32), g["kind"]
    n = len(a)
    ref = box_ref.iou_pair(box_ref.boxes3d2corners(a.astype(np.float64)), box_ref.boxes3d2corners(b.astype(np.float64)))
    out = np.zeros((n, 2), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    host_lib.host_iou_from_params(a.ctypes.data_as(fp), b.ctypes.data_as(fp), n, out.ctypes.data_as(fp))
    assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()
    ca = box_ref.boxes3d2corners(a).astype(np.float32).copy()
    cb = box_ref.boxes3d2corners(b).astype(np.float32).copy()
    out2 = np.zeros((n, 2), dtype=np.float32)
    host_lib.host_iou_from_corners(ca.ctypes.data_as(fp), cb.ctypes.data_as(fp), n, out2.ctypes.data_as(fp))
    assert np.abs(out2 - ref).max() < 5e-5, np.abs(out2 - ref).max()


@pytest.mark.parametrize("variant", ["full", "plain", "allbg"])
@pytest.mark.parametrize("method", ["nms", "top"])
def test_decode_matches_reference_test_loop(variant, method):
    """oracle/box_ref.decode_detections against the rows the reference's own test() loop (train/test_net_det.py:193-293, run
    by tests/golden/make_golden_decode.py with the reference model's eval outputs) produced: foreground selection with its
    arg-max fallback, score = p_fg + rgb_prob, from_prediction_to_label_format, the too-small filter, row order."""
    g = load_golden("decode_b6_n512")
    probs = g["allbg_cls_probs"] if variant == "allbg" else g["eval_cls_probs"]
    B = probs.shape[0]
    rows, counts = g["rows_%s_%s" % (variant, method)], g["counts_%s_%s" % (variant, method)]
    extras = variant != "plain"
    off = 0
    for b in range(B):
        refc = g["ref_center"][b].astype(np.float64) if extras else np.zeros(3)       # test_net_det.py:205-211 defaults
        rgb = float(g["rgb_prob"][b, 0]) if extras else 1.0
        got, _ = box_ref.decode_detections(probs[b].astype(np.float64), g["eval_center"][b].astype(np.float64),
                                           g["eval_heading"][b].astype(np.float64), g["eval_size"][b].astype(np.float64),
                                           float(g["rot_angle"][b, 0]), refc, rgb, method)
        exp = rows[off:off + counts[b]][:, [4, 5, 6, 9, 8, 7, 10, 11]]      # (tx,ty,tz,h,w,l,ry,score) -> (tx,ty,tz,l,w,h,ry,score)
        off += counts[b]
        assert got.shape == exp.shape, (b, got.shape, exp.shape)
        # the reference computes these in float32 (its rows are float32 values): one float32 ulp of slack
        assert (np.abs(got - exp) <= 1e-6 + 2.5e-7 * np.abs(exp)).all(), (b, np.abs(got - exp).max())
    assert off == len(rows) and (counts > 0).all()
