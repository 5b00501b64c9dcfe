"""ORACLE -- test infrastructure only (see oracle/__init__.py).

Numpy restatement of the reference's rotated-box overlap path:
  * boxes3d2corners           ops/pybind11/rbbox_iou.py:121-148 (== models/model_util.py:48-72 get_box3d_corners_helper)
  * iou_pair                  rbbox_iou_3d_pair, ops/pybind11/box_ops.h:173-260 (BEV polygon = corners 6,7,4,5; y extents
                              from corners 0 / 4), with the polygon clipping of utils/box_util.py:11-56 (Sutherland-Hodgman)
  * cube_nms                  rotate_nms_3d_cc, ops/pybind11/rbbox_iou.py:294-311 -> rotate_non_max_suppression_3d_cpu,
                              ops/pybind11/nms_cpu.h:148-240 (greedy by descending score, suppress when IoU3D >= thresh)
  * decode_detections         train/test_net_det.py:254-293 (foreground selection, score) +
                              from_prediction_to_label_format, datasets/provider_sample.py:375-387
  * iou_metrics               the no_grad metric block of models/det_base.py:480-503

Pinning: box_ops_cc / nms need boost::geometry (un-vendored, absent) -> unbuildable here, so there is no oracle/_ref.
iou_pair is pinned by tests/golden/box_iou_pairs.npz = outputs of the reference's own pure-python
utils/box_util.py:121-150 (box3d_iou_pair) run here by tests/golden/make_golden_iou.py; cube_nms by keep lists that the
reference's cube_nms_np loop (rbbox_iou.py:203-236) produced with that same python IoU injected for the boost module.
"""
import numpy as np


def boxes3d2corners(boxes):
    """(n,7) (cx,cy,cz,l,w,h,ry) -> (n,8,3) corners in the reference's order."""
    boxes = np.asarray(boxes)
    l, w, h, r = boxes[:, 3], boxes[:, 4], boxes[:, 5], boxes[:, 6]
    xs = np.stack([l, l, -l, -l, l, l, -l, -l], 1) / 2
    ys = np.stack([h, h, h, h, -h, -h, -h, -h], 1) / 2
    zs = np.stack([w, -w, -w, w, w, -w, -w, w], 1) / 2
    c, s = np.cos(r)[:, None], np.sin(r)[:, None]
    x = c * xs + s * zs + boxes[:, 0:1]
    y = ys + boxes[:, 1:2]
    z = -s * xs + c * zs + boxes[:, 2:3]
    return np.stack([x, y, z], 2).astype(boxes.dtype)


def _clip(subject, clip):
    """Sutherland-Hodgman (utils/box_util.py:11-56), clip polygon in either orientation."""
    cx, cz = clip[:, 0], clip[:, 1]
    sgn = 1.0 if (np.dot(cx, np.roll(cz, -1)) - np.dot(cz, np.roll(cx, -1))) >= 0 else -1.0
    out = [tuple(p) for p in subject]
    for e in range(len(clip)):
        c1, c2 = clip[e], clip[(e + 1) % len(clip)]
        ex, ez = c2[0] - c1[0], c2[1] - c1[1]
        inp, out = out, []
        if not inp:
            return []
        s = inp[-1]
        ds = sgn * (ex * (s[1] - c1[1]) - ez * (s[0] - c1[0]))
        for v in inp:
            dv = sgn * (ex * (v[1] - c1[1]) - ez * (v[0] - c1[0]))
            if (dv > 0) != (ds > 0):
                t = ds / (ds - dv)
                out.append((s[0] + t * (v[0] - s[0]), s[1] + t * (v[1] - s[1])))
            if dv > 0:
                out.append(v)
            s, ds = v, dv
    return out


def _area(poly):
    if len(poly) < 3:
        return 0.0
    p = np.asarray(poly, dtype=np.float64)
    return 0.5 * abs(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))


def iou_pair(corners1, corners2):
    """(n,8,3) x 2 -> (n,2) [BEV IoU, 3-D IoU] (float64 arithmetic)."""
    c1 = np.asarray(corners1, dtype=np.float64)
    c2 = np.asarray(corners2, dtype=np.float64)
    out = np.zeros((len(c1), 2))
    for i in range(len(c1)):
        a = c1[i][[6, 7, 4, 5]][:, [0, 2]]
        b = c2[i][[6, 7, 4, 5]][:, [0, 2]]
        inter = _area(_clip(a, b))
        if inter <= 0:
            continue
        aa, ab = _area(a), _area(b)
        ymax = min(c1[i, 0, 1], c2[i, 0, 1])
        ymin = max(c1[i, 4, 1], c2[i, 4, 1])
        ivol = inter * max(0.0, ymax - ymin)
        va = max(0.0, aa * (c1[i, 0, 1] - c1[i, 4, 1]))
        vb = max(0.0, ab * (c2[i, 0, 1] - c2[i, 4, 1]))
        out[i, 0] = inter / (aa + ab - inter)
        out[i, 1] = ivol / (va + vb - ivol)
    return out


def cube_nms(dets, thresh, top_k=300):
    """dets (n,8) [cx,cy,cz,l,w,h,ry,score] -> kept indices in keep order."""
    dets = np.asarray(dets)
    n = len(dets)
    if n == 0:
        return []
    order = dets[:, 7].argsort()[::-1]
    corners = boxes3d2corners(dets[:, :7])
    lo, hi = corners.min(1), corners.max(1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            if np.any(np.minimum(hi[i], hi[j]) - np.maximum(lo[i], lo[j]) <= 0):      # standup_iou <= 0
                continue
            if iou_pair(corners[i:i + 1], corners[j:j + 1])[0, 1] >= thresh:
                suppressed[j] = True
    return keep[:top_k]


def rotate_y(xyz, ang):
    """rotate_pc_along_y (datasets/provider_sample.py:329-343): x' = c x - s z, z' = s x + c z."""
    c, s = np.cos(ang), np.sin(ang)
    x, y, z = xyz
    return np.array([c * x - s * z, y, s * x + c * z])


def decode_detections(cls_probs, center_preds, heading_preds, size_preds, rot_angle, ref_center, rgb_prob, method="nms"):
    """One frustum of train/test_net_det.py:254-293: -> (rows (m,8) [tx,ty,tz,l,w,h,ry,score] in label format, fg indices)."""
    if method == "nms":
        fg = np.nonzero(cls_probs[:, 0] < cls_probs[:, 1])[0]
        if fg.size == 0:
            fg = np.array([np.argmax(cls_probs[:, 1])])
    else:
        fg = np.array([np.argmax(cls_probs[:, 1])])
    rows, idx = [], []
    for n in fg:
        l, w, h = size_preds[n]
        ry = heading_preds[n] + rot_angle
        tx, ty, tz = rotate_y(center_preds[n], -rot_angle) + ref_center
        ty = ty + h / 2.0
        if h < 0.01 or w < 0.01 or l < 0.01:
            continue
        rows.append([tx, ty, tz, l, w, h, ry, cls_probs[n, 1] + rgb_prob])
        idx.append(int(n))
    return np.array(rows, dtype=np.float64).reshape(-1, 8), idx


def iou_metrics(reg_rows, center_ref2_rows, fg_rows, box3d_center, box3d_heading, box3d_size, mean_size, nb=12, ns=3,
                thresh=0.7):
    """The no_grad metric block of models/det_base.py:480-503 on the foreground rows:
    reg_rows (R, 3+2nb+4ns), center_ref2_rows (R,3), fg_rows (list of (row, frustum)) -> (IoU_2D, IoU_3D, IoU_>=thresh)."""
    per = 2 * np.pi / nb
    pred, gt = [], []
    for r, b in fg_rows:
        o = np.asarray(reg_rows[r], dtype=np.float64)
        ah = int(np.argmax(o[3:3 + nb]))
        a_s = int(np.argmax(o[3 + 2 * nb:3 + 2 * nb + ns]))
        ang = ah * per + o[3 + nb + ah] * (per / 2)
        if ang > np.pi:
            ang -= 2 * np.pi
        sr = o[3 + 2 * nb + ns + 3 * a_s:3 + 2 * nb + ns + 3 * a_s + 3]
        size = sr * mean_size[a_s] + mean_size[a_s]
        c = o[0:3] + center_ref2_rows[r]
        pred.append(np.concatenate([c, size, [ang]]))
        gt.append(np.concatenate([box3d_center[b], box3d_size[b], [float(box3d_heading[b])]]))
    if not pred:
        return 0.0, 0.0, 0.0
    ov = iou_pair(boxes3d2corners(np.array(pred)), boxes3d2corners(np.array(gt)))
    return float(ov[:, 0].mean()), float(ov[:, 1].mean()), float((ov[:, 1] >= thresh).mean())
