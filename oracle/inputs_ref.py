"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's per-sample input construction, the row
"next-1" of SURVEY.md section 8f: datasets/provider_sample.py::ProviderDataset.__getitem__ (:137-262) with
generate_ref (:291-327), generate_labels (:270-289) and the centre-view helpers (:329-372), on top of
datasets/data_utils.py (rotate_pc_along_y :7-21, compute_box_3d :44-70, project_image_to_rect :73-93).

Pure numpy, fp64 where the reference computes in fp64, rounded to fp32 where it stores fp32.  Randomness is an INPUT:
the three draws the reference takes per sample (np.random.choice resample, np.random.random flip coin,
np.random.randn shift) are passed in, so the restatement is deterministic and comparable.

Pinned by tests/golden/inputs_kitti_b6.npz: outputs of the reference's own ProviderDataset run on a synthetic pickle with
its RNG calls recorded (tests/golden/make_golden_inputs.py; tests/test_oracle_inputs.py).  One deliberate difference:
the reference decides "centre inside the (half) box" with scipy's Delaunay hull (data_utils.py:24-37); here it is the
closed-form oriented-box test, identical except for points within qhull's ~1e-12 tolerance of a face.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


def rotate_along_y(xz, rot_angle):
    """(n,2) columns (x, z) -> rotated, fp64 (data_utils.py:16-20: pc[:, [0,2]] . rotmat^T)."""
    c, s = np.cos(rot_angle), np.sin(rot_angle)
    x, z = xz[:, 0].astype(np.float64), xz[:, 1].astype(np.float64)
    return np.stack([x * c + z * (-s), x * s + z * c], 1)


def generate_ref(box2d, P, strides, max_depth):
    """provider_sample.py:291-327 + data_utils.py:73-93: frustum centres on the ray through the 2-D box centre."""
    cu, cv, fu, fv = P[0, 2], P[1, 2], P[0, 0], P[1, 1]
    bx, by = P[0, 3] / (-fu), P[1, 3] / (-fv)
    cx, cy = (box2d[0] + box2d[2]) / 2.0, (box2d[1] + box2d[3]) / 2.0
    refs = []
    for s in strides:
        z = np.arange(0, max_depth, s) + s / 2.0
        x = ((cx - cu) * z) / fu + bx
        y = ((cy - cv) * z) / fv + by
        refs.append(np.stack([x, y, z], 1))
    return refs


def in_box(p, center, dims, angle):
    """Closed-form stand-in for extract_pc_in_box3d(p, compute_box_3d(center, dims, angle)) (data_utils.py:31-70):
    corners = roty(angle) . (+-l/2, +-h/2, +-w/2) + center  =>  local = roty(angle)^T (p - center)."""
    c, s = np.cos(angle), np.sin(angle)
    d = p - center[None, :]
    lx = c * d[:, 0] - s * d[:, 2]
    lz = s * d[:, 0] + c * d[:, 2]
    l, w, h = dims
    return (np.abs(lx) <= l / 2.0) & (np.abs(d[:, 1]) <= h / 2.0) & (np.abs(lz) <= w / 2.0)


def generate_labels(center, size, angle, ref):
    """provider_sample.py:270-289: 1 inside the half-size box, -1 inside the full box, else 0; nearest centre = 1
    when no centre is inside the half-size box."""
    labels = np.zeros(len(ref), dtype=np.int64)
    inside1 = in_box(ref, center, size * 0.5, angle)
    inside2 = in_box(ref, center, size, angle)
    labels[inside2] = -1
    labels[inside1] = 1
    if inside1.sum() == 0:
        dis = np.sqrt(((ref - center[None, :]) ** 2).sum(1))
        labels[np.argmin(dis)] = 1
    return labels


def prepare_sample(raw_pts, raw_seg, box2d, P, box3d_corners, heading, size, frustum_angle, choice, coin, normal,
                   strides, max_depth, random_flip=True, random_shift=True):
    """One training sample with rotate-to-centre (cfg.DATA.RTC), provider_sample.py:137-262."""
    rot = np.pi / 2.0 + frustum_angle                                           # :329-332
    xz = rotate_along_y(raw_pts[:, [0, 2]], rot).astype(np.float32)             # stored back into the float32 record
    pts = np.stack([xz[:, 0], raw_pts[:, 1].astype(np.float32), xz[:, 1]], 1)[choice]   # :155-170
    seg = raw_seg[choice]
    refs = generate_ref(box2d, P, strides, max_depth)
    for r in refs:                                                              # :176-180
        r[:, [0, 2]] = rotate_along_y(r[:, [0, 2]], rot)
    c0 = (box3d_corners[0] + box3d_corners[6]) / 2.0                            # :339-346
    cxz = rotate_along_y(c0[None, [0, 2]], rot)[0]
    center = np.array([cxz[0], c0[1], cxz[1]], dtype=np.float64)
    angle = heading - rot                                                       # :214-218
    if random_flip and coin > 0.5:                                              # :222-233
        pts[:, 0] *= -1
        center[0] *= -1
        angle = np.pi - angle
        for r in refs:
            r[:, 0] *= -1
    if random_shift:                                                            # :235-242
        l, w, h = size
        dist = np.sqrt(np.sum(l ** 2 + w ** 2))
        shift = np.clip(normal * dist * 0.2, -0.5 * dist, 0.5 * dist)
        shift = np.clip(shift + center[2], 0, max_depth) - center[2]
        pts[:, 2] = (pts[:, 2].astype(np.float64) + shift).astype(np.float32)   # float32 record += float64 scalar
        center[2] += shift
    labels = generate_labels(center, np.asarray(size, dtype=np.float64), angle, refs[1])
    out = {"point_cloud": np.ascontiguousarray(pts.T), "rot_angle": np.array([rot], dtype=np.float32),
           "cls_label": labels, "box3d_center": center.astype(np.float32),
           "box3d_heading": np.array([angle], dtype=np.float32), "box3d_size": np.asarray(size).astype(np.float32),
           "seg_label": seg.astype(np.int64)}
    for i, r in enumerate(refs):
        out["center_ref%d" % (i + 1)] = np.ascontiguousarray(r.astype(np.float32).T)
    return out


def prepare_batch(rec, strides, max_depth, random_flip=True, random_shift=True):
    """rec: dict with raw_points (sum n, C) float32, raw_seg, raw_counts, box2d, P, box3d_corners, heading, size,
    frustum_angle, draw_choice, draw_coin, draw_normal (the fixture's layout).  Returns the collated batch."""
    offs = np.concatenate([[0], np.cumsum(rec["raw_counts"])])
    items = []
    for b in range(len(rec["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        items.append(prepare_sample(rec["raw_points"][sl], rec["raw_seg"][sl], rec["box2d"][b], rec["P"][b],
                                    rec["box3d_corners"][b], float(rec["heading"][b]), rec["size"][b],
                                    float(rec["frustum_angle"][b]), rec["draw_choice"][b], float(rec["draw_coin"][b]),
                                    float(rec["draw_normal"][b]), strides, max_depth, random_flip, random_shift))
    return {k: np.stack([it[k] for it in items]) for k in items[0]}


# ------------------------------------------------------------------------------------------------
# Refinement stage (cfgs/refine_car.yaml): datasets/provider_sample_refine.py::ProviderDataset.__getitem__ (:176-315) with
# generate_ref (:336-386), generate_labels (:317-334), get_center_view_* (:137-152) and collate_fn (:388-419: per-sample L
# differs, center_ref* / cls_label edge-padded to the batch maximum).  Pinned by tests/golden/inputs_refine_b6.npz = outputs
# of the reference's own refine ProviderDataset + collate_fn (tests/golden/make_golden_inputs_refine.py).

def box_corners(center, dims, angle):
    """datasets/data_utils.py:44-70 compute_box_3d."""
    l, w, h = dims
    c, s = np.cos(angle), np.sin(angle)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    xs = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    ys = [h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2]
    zs = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    cor = np.dot(R, np.vstack([xs, ys, zs]))
    cor[0] += center[0]; cor[1] += center[1]; cor[2] += center[2]
    return cor.T


def generate_ref_refine(pred_box3d, strides):
    """provider_sample_refine.py:336-386: centres along the line through the front / back face centres of the box, over its
    z extent."""
    cz = ((pred_box3d[0] + pred_box3d[6]) / 2)[2]
    z1, z2 = np.min(pred_box3d[:, 2]), np.max(pred_box3d[:, 2])
    c1 = np.mean(pred_box3d[pred_box3d[:, 2] < cz], 0)
    c2 = np.mean(pred_box3d[pred_box3d[:, 2] > cz], 0)
    delta = c2 - c1
    refs = []
    for s in strides:
        z = np.arange(z1, z2, s) + s / 2.0
        x = (z - c1[2]) / delta[2] * delta[0] + c1[0]
        y = (z - c1[2]) / delta[2] * delta[1] + c1[1]
        refs.append(np.stack([x, y, z], 1))
    return refs


def generate_labels_refine(center, size, angle, ref):
    """provider_sample_refine.py:317-334: boxes scaled by 0.3 (positive) / 0.6 (ignore)."""
    labels = np.zeros(len(ref), dtype=np.int64)
    inside1 = in_box(ref, center, size * 0.3, angle)
    inside2 = in_box(ref, center, size * 0.6, angle)
    labels[inside2] = -1
    labels[inside1] = 1
    if inside1.sum() == 0:
        dis = np.sqrt(((ref - center[None, :]) ** 2).sum(1))
        labels[np.argmin(dis)] = 1
    return labels


def prepare_sample_refine(raw_pts, pred_corners, pred_angle, pred_size, gt_corners, gt_heading, gt_size, choice, coin,
                          normal, strides, random_flip=True, random_shift=True):
    pc = (pred_corners[0] + pred_corners[6]) / 2.0                               # :181
    d = raw_pts[:, :3].astype(np.float64) - pc[None, :]                          # get_center_view_point :146-152
    xz = rotate_along_y(d[:, [0, 2]], pred_angle)
    pts = np.stack([xz[:, 0], d[:, 1], xz[:, 1]], 1).astype(np.float32)[choice]  # stored back into the float32 record
    zero = rotate_along_y(np.zeros((1, 2)), pred_angle)[0]                       # centre view of the predicted box itself
    pred_box = box_corners(np.array([zero[0], 0.0, zero[1]]), pred_size, 0.0)    # :219-228
    refs = generate_ref_refine(pred_box, strides)
    c0 = (gt_corners[0] + gt_corners[6]) / 2.0 - pc                              # get_center_view_box3d :137-144
    cxz = rotate_along_y(c0[None, [0, 2]], pred_angle)[0]
    center = np.array([cxz[0], c0[1], cxz[1]], dtype=np.float64)
    angle = gt_heading - pred_angle
    if random_flip and coin > 0.5:                                               # :265-276
        pts[:, 0] *= -1
        center[0] *= -1
        angle = np.pi - angle
        for r in refs:
            r[:, 0] *= -1
    if random_shift:                                                             # :278-284
        l, w, h = gt_size
        dist = np.sqrt(np.sum(l ** 2 + w ** 2))
        s1 = strides[0]
        shift = np.clip(normal * dist * 0.1, -s1 * 2, 2 * s1)
        pts[:, 2] = (pts[:, 2].astype(np.float64) + shift).astype(np.float32)
        center[2] += shift
    labels = generate_labels_refine(center, np.asarray(gt_size, dtype=np.float64), angle, refs[1])
    out = {"point_cloud": np.ascontiguousarray(pts.T), "cls_label": labels, "box3d_center": center.astype(np.float32),
           "box3d_heading": np.array([angle], dtype=np.float32), "box3d_size": np.asarray(gt_size).astype(np.float32),
           "rot_angle": np.array([pred_angle], dtype=np.float32), "ref_center": pc.astype(np.float32)}
    for i, r in enumerate(refs):
        out["center_ref%d" % (i + 1)] = np.ascontiguousarray(r.astype(np.float32).T)
    return out


def collate_refine(items):
    """provider_sample_refine.py:388-419: edge-pad the variable-length keys to the batch maximum, then stack."""
    names = ["center_ref1", "center_ref2", "center_ref3", "center_ref4", "cls_label"]
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        if k in names:
            m = max(v.shape[-1] for v in vals)
            vals = [np.pad(v, [(0, 0)] * (v.ndim - 1) + [(0, m - v.shape[-1])], mode="edge") for v in vals]
        out[k] = np.stack(vals)
    return out


def prepare_batch_refine(rec, strides, random_flip=True, random_shift=True):
    offs = np.concatenate([[0], np.cumsum(rec["raw_counts"])])
    items = []
    for b in range(len(rec["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        items.append(prepare_sample_refine(rec["raw_points"][sl], rec["pred_corners"][b], float(rec["pred_angle"][b]),
                                           rec["pred_size"][b], rec["box3d_corners"][b], float(rec["heading"][b]),
                                           rec["size"][b], rec["draw_choice"][b], float(rec["draw_coin"][b]),
                                           float(rec["draw_normal"][b]), strides, random_flip, random_shift))
    return collate_refine(items)


# ------------------------------------------------------------------------------------------------
# SUN-RGBD loader (cfgs/det_sample_sunrgbd.yaml): datasets/provider_sample_sunrgbd.py::ProviderDataset.__getitem__
# (:116-263) with generate_ref (:283-326), project_image_to_camera / _to_upright_camera (:28-59), generate_labels (:265-281).
# Differences from the KITTI loader above: five strides; the centres go through K and Rtilt; np.random.choice replaces only
# when the frustum has fewer points than npoints (:144); random_shift draws a height shift too (:228-230).  Pinned by
# tests/golden/inputs_sunrgbd_b6.npz = outputs of the reference's own ProviderDataset (make_golden_inputs_sunrgbd.py).

def project_image_to_upright_camera(uv_depth, K, Rtilt):
    """provider_sample_sunrgbd.py:28-59."""
    c_u, c_v, f_u, f_v = K[0, 2], K[1, 2], K[0, 0], K[1, 1]
    x = ((uv_depth[:, 0] - c_u) * uv_depth[:, 2]) / f_u
    y = ((uv_depth[:, 1] - c_v) * uv_depth[:, 2]) / f_v
    depth = np.stack([x, uv_depth[:, 2], -y], 1)                    # X, Y, Z -> X, Z, -Y
    up = (Rtilt @ depth.T).T
    return np.stack([up[:, 0], -up[:, 2], up[:, 1]], 1)             # X, Y, Z -> X, -Z, Y


def generate_ref_sunrgbd(box2d, K, Rtilt, strides, max_depth):
    """provider_sample_sunrgbd.py:283-326."""
    cx, cy = (box2d[0] + box2d[2]) / 2.0, (box2d[1] + box2d[3]) / 2.0
    refs = []
    for s in strides:
        z = np.arange(0, max_depth, s) + s / 2.0
        uvd = np.stack([np.full_like(z, cx), np.full_like(z, cy), z], 1)
        refs.append(project_image_to_upright_camera(uvd, K, Rtilt))
    return refs


def prepare_sample_sunrgbd(raw_pts, raw_seg, box2d, K, Rtilt, box3d_corners, heading, size, frustum_angle, choice, coin,
                           normal, hshift, strides, max_depth, random_flip=True, random_shift=True):
    """One training sample with rotate-to-centre, provider_sample_sunrgbd.py:116-263.  hshift: the np.random.random() draw
    behind the height shift (:228)."""
    rot = np.pi / 2.0 + frustum_angle                                           # :328-331
    xz = rotate_along_y(raw_pts[:, [0, 2]], rot).astype(raw_pts.dtype)          # stored back into the record's dtype
    pts = np.stack([xz[:, 0], raw_pts[:, 1], xz[:, 1]], 1)[choice]              # :133-150
    seg = raw_seg[choice]
    refs = generate_ref_sunrgbd(box2d, K, Rtilt, strides, max_depth)
    for r in refs:                                                              # :158-163
        r[:, [0, 2]] = rotate_along_y(r[:, [0, 2]], rot)
    c0 = (box3d_corners[0] + box3d_corners[6]) / 2.0                            # :339-344
    cxz = rotate_along_y(c0[None, [0, 2]], rot)[0]
    center = np.array([cxz[0], c0[1], cxz[1]], dtype=np.float64)
    angle = heading - rot                                                       # :198-201
    if random_flip and coin > 0.5:                                              # :208-220
        pts[:, 0] *= -1
        center[0] *= -1
        angle = np.pi - angle
        for r in refs:
            r[:, 0] *= -1
    if random_shift:                                                            # :222-230
        l, w, h = size
        dist = np.sqrt(np.sum(l ** 2 + w ** 2))
        shift = np.clip(normal * dist * 0.2, -0.5 * dist, 0.5 * dist)
        shift = np.clip(shift + center[2], 0, max_depth) - center[2]
        pts[:, 2] = (pts[:, 2].astype(np.float64) + shift).astype(pts.dtype)
        center[2] += shift
        height_shift = hshift * 0.4 - 0.2
        pts[:, 1] = (pts[:, 1].astype(np.float64) + height_shift).astype(pts.dtype)
        center[1] += height_shift
    labels = generate_labels(center, np.asarray(size, dtype=np.float64), angle, refs[1])     # :232, :265-281
    out = {"point_cloud": np.ascontiguousarray(pts.astype(np.float32).T), "rot_angle": np.array([rot], dtype=np.float32),
           "cls_label": labels, "box3d_center": center.astype(np.float32),
           "box3d_heading": np.array([angle], dtype=np.float32), "box3d_size": np.asarray(size).astype(np.float32),
           "seg_label": seg.astype(np.int64)}
    for i, r in enumerate(refs):
        out["center_ref%d" % (i + 1)] = np.ascontiguousarray(r.astype(np.float32).T)
    return out


def prepare_batch_sunrgbd(rec, strides, max_depth, random_flip=True, random_shift=True):
    """rec: the fixture's layout (raw_points, raw_seg, raw_counts, box2d, K, Rtilt, box3d_corners, heading, size,
    frustum_angle, draw_choice, draw_coin, draw_normal, draw_hshift).  Returns the collated batch."""
    offs = np.concatenate([[0], np.cumsum(rec["raw_counts"])])
    items = []
    for b in range(len(rec["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        items.append(prepare_sample_sunrgbd(rec["raw_points"][sl], rec["raw_seg"][sl], rec["box2d"][b], rec["K"][b],
                                            rec["Rtilt"][b], rec["box3d_corners"][b], float(rec["heading"][b]),
                                            rec["size"][b], float(rec["frustum_angle"][b]), rec["draw_choice"][b],
                                            float(rec["draw_coin"][b]), float(rec["draw_normal"][b]),
                                            float(rec["draw_hshift"][b]), strides, max_depth, random_flip, random_shift))
    return {k: np.stack([it[k] for it in items]) for k in items[0]}
