"""Deterministic synthetic frustum batches and counter-hash weights.

Everything here is a pure function of integer seeds: a splitmix64 counter hash
turned into uniforms with exact float64 arithmetic (no libm calls), so the GPU
box, the CPU oracle and the fixture generator all see bit-identical inputs
without depending on torch's RNG.

Shapes follow the reference's data provider (datasets/provider_sample.py:248-262
for the dict keys, :291-299 for the sliding-frustum centres).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
CAR_MEAN_SIZE = (3.88311640418, 1.62856739989, 1.52563191462)


def _splitmix(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x = x * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def stream_id(name):
    """Stable 32-bit id of a tensor / stream name."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def uniform01(seed, stream, shape, lane=0):
    """float64 uniforms in [0,1) of `shape`; element i of stream (seed, stream, lane)."""
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64)
    key = _splitmix(np.array([(int(seed) << 32) ^ int(stream)], dtype=np.uint64))
    key = _splitmix(key ^ np.uint64(int(lane) * 0x51ED27 + 1))
    with np.errstate(over="ignore"):
        h = _splitmix(ctr * np.uint64(0xD1342543DE82EF95) + key)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normalish(seed, stream, shape):
    """Zero-mean unit-variance samples (Irwin-Hall of 12 uniforms): exact arithmetic only."""
    acc = np.zeros(shape, dtype=np.float64)
    for lane in range(12):
        acc += uniform01(seed, stream, shape, lane=lane + 1)
    return acc - 6.0


def frustum_centres(stride, max_depth=70.0):
    """z of the sliding-frustum centres for one scale (provider_sample.py:293-299)."""
    return np.arange(0, max_depth, stride) + stride / 2.0


def make_batch(batch, npoint, strides=(0.25, 0.5, 1.0, 2.0), max_depth=70.0, seed=1234,
               variant="car", tilt=(0.0, 0.0), with_labels=True, z_range=None, num_classes=3, mean_sizes=None):
    """Synthetic KITTI-car-shaped batch as a dict of numpy arrays (reference dict keys).

    variant: "car"     60 % foreground z ~ N(z_obj, 0.8^2), 40 % background U(0, max_depth)
             "uniform" z ~ U(0, max_depth)
    tilt:    (kx, ky) centre x,y = k * z (the reference projects the 2-D box centre along depth)
    z_range: optional (lo, hi) to confine everything (refine-stage style short frustums)
    num_classes / mean_sizes: class count of the one-hot vector and the (num_classes, 3) mean-size table.  The defaults
             give the KITTI batch (class 0 = Car for every sample); with a table (SUN-RGBD: 10 classes) sample b is an
             object of class b % num_classes whose size is drawn around that class's mean.
    """
    B, N = batch, npoint
    lo, hi = (0.0, float(max_depth)) if z_range is None else z_range
    u = lambda name, shape, lane=0: uniform01(seed, stream_id(name), shape, lane)
    z_obj = lo + (hi - lo) * (5.0 / 70.0 + u("z_obj", (B,)) * (55.0 / 70.0))
    z_bg = lo + (hi - lo) * u("z_bg", (B, N))
    if variant == "car":
        is_fg = u("fg_mask", (B, N)) < 0.6
        z_fg = z_obj[:, None] + 0.8 * normalish(seed, stream_id("z_fg"), (B, N))
        z = np.where(is_fg, z_fg, z_bg)
    elif variant == "uniform":
        z = z_bg
    else:
        raise ValueError(variant)
    eps = 0.05 * (hi - lo) / 70.0
    z = np.clip(z, lo + eps, hi - eps)
    x = (u("x", (B, N)) * 4.0 - 2.0) * (np.abs(z) / 20.0 + 0.2)
    y = u("y", (B, N)) * 2.5 - 1.5
    pc = np.stack([x, y, z], axis=1).astype(np.float32)  # (B,3,N)

    out = {"point_cloud": pc}
    obj_cls = np.zeros(B, dtype=np.int64) if mean_sizes is None else np.arange(B, dtype=np.int64) % num_classes
    one_hot = np.zeros((B, num_classes), dtype=np.float32)
    one_hot[np.arange(B), obj_cls] = 1.0
    out["one_hot"] = one_hot
    for s, stride in enumerate(strides):
        if z_range is None:
            zc = frustum_centres(stride, max_depth)
        else:
            zc = np.arange(lo, hi, stride) + stride / 2.0
        ref = np.zeros((B, 3, len(zc)), dtype=np.float64)
        ref[:, 0, :] = tilt[0] * zc
        ref[:, 1, :] = tilt[1] * zc
        ref[:, 2, :] = zc
        out["center_ref%d" % (s + 1)] = ref.astype(np.float32)
    if with_labels:
        zc2 = out["center_ref2"][0, 2].astype(np.float64)
        L2 = len(zc2)
        nearest = np.argmin(np.abs(zc2[None, :] - z_obj[:, None]), axis=1)
        cls = np.zeros((B, L2), dtype=np.int64)
        for b in range(B):
            c = int(nearest[b])
            if c - 1 >= 0:
                cls[b, c - 1] = -1
            if c + 1 < L2:
                cls[b, c + 1] = -1
            cls[b, c] = 1
        out["cls_label"] = cls
        out["size_class"] = obj_cls.reshape(B, 1).copy()
        ctr = np.zeros((B, 3), dtype=np.float64)
        ctr[:, 0] = tilt[0] * z_obj
        ctr[:, 1] = tilt[1] * z_obj
        ctr[:, 2] = z_obj
        out["box3d_center"] = ctr.astype(np.float32)
        out["box3d_heading"] = ((u("heading", (B, 1)) * 2.0 - 1.0) * np.pi).astype(np.float32)
        base = np.array(CAR_MEAN_SIZE)[None, :] if mean_sizes is None else np.asarray(mean_sizes, dtype=np.float64)[obj_cls]
        size = base * (0.9 + 0.2 * u("size", (B, 3)))
        out["box3d_size"] = size.astype(np.float32)
    return out


KITTI_P2 = ((721.5377, 0.0, 609.5593, 44.85728), (0.0, 721.5377, 172.854, 0.2163791), (0.0, 0.0, 1.0, 0.002745884))


def make_records(batch, seed=1234, n_raw=(1200, 2400), max_depth=70.0):
    """Synthetic RAW frustum records -- what the reference's loader reads from its pickles before __getitem__ builds a sample
    (datasets/provider_sample.py:137-262; keys = frustum_convnet_amd.inputs.RECORD_KEYS): per frustum a KITTI-car-shaped point
    set in RECT CAMERA coordinates (n, 4) float32 (n ~ U(n_raw): the loader resamples it to NUM_SAMPLES), its segmentation mask,
    the 2-D box and KITTI projection matrix whose centre ray carries the sliding-frustum centres, the 3-D box (corners, heading,
    size) of a car on that ray, and the frustum angle (-atan2(z, x) of the box-centre ray).  Same point statistics as make_batch
    (60 % foreground around the object, 40 % background along the ray), laid out in the frustum's own frame and rotated back."""
    P = np.asarray(KITTI_P2, dtype=np.float64)
    fu, fv, cu, cv = P[0, 0], P[1, 1], P[0, 2], P[1, 2]
    bx, by = P[0, 3] / (-fu), P[1, 3] / (-fv)
    u = lambda name, shape, lane=0: uniform01(seed, stream_id(name), shape, lane)
    z_obj = 5.0 + u("rec_z_obj", (batch,)) * 55.0
    px = 200.0 + u("rec_px", (batch,)) * 840.0               # 2-D box centre (pixels)
    py = 150.0 + u("rec_py", (batch,)) * 60.0
    nraw = (n_raw[0] + u("rec_n", (batch,)) * (n_raw[1] - n_raw[0])).astype(np.int64)
    heading = (u("rec_heading", (batch,)) * 2.0 - 1.0) * np.pi
    size = np.array(CAR_MEAN_SIZE)[None, :] * (0.9 + 0.2 * u("rec_size", (batch, 3)))
    recs = []
    for b in range(batch):
        ray = lambda z: np.stack([(px[b] - cu) * z / fu + bx, (py[b] - cv) * z / fv + by, z], -1)      # (data_utils.py:73-93)
        c = ray(np.array([20.0]))[0]                       # (the reference takes the angle of the ray's point at depth 20)
        fangle = -np.arctan2(c[2], c[0])
        rot = np.pi / 2.0 + fangle
        cs, sn = np.cos(rot), np.sin(rot)
        ctr0 = ray(np.array([z_obj[b]]))[0]
        d_obj = ctr0[0] * sn + ctr0[2] * cs                # the object's depth in the frustum's own (rotated) frame
        n = int(nraw[b])
        lane = 1000 * (b + 1)
        is_fg = u("rec_fg", (n,), lane) < 0.6
        zf = d_obj + 0.8 * normalish(seed + b, stream_id("rec_zfg"), (n,))
        zb = u("rec_zbg", (n,), lane) * max_depth
        d = np.clip(np.where(is_fg, zf, zb), 0.05, max_depth - 0.05)
        # frustum frame (x lateral, y vertical, d depth along the centre ray) -> rect camera frame: the inverse of rotate_pc_along_y
        xl = (u("rec_x", (n,), lane) * 4.0 - 2.0) * (d / 20.0 + 0.2)
        yl = u("rec_y", (n,), lane) * 2.5 - 1.5
        xc = xl * cs + d * sn
        zc = -xl * sn + d * cs
        pts = np.stack([xc, (py[b] - cv) * zc / fv + by + yl, zc, u("rec_i", (n,), lane)], 1).astype(np.float32)
        ctr = ray(np.array([z_obj[b]]))[0]
        l, w, h = size[b]
        ch, sh = np.cos(heading[b]), np.sin(heading[b])
        xs = np.array([l, l, -l, -l, l, l, -l, -l]) / 2.0
        ys = np.array([h, h, h, h, -h, -h, -h, -h]) / 2.0
        zs = np.array([w, -w, -w, w, w, -w, -w, w]) / 2.0
        corners = np.stack([ch * xs + sh * zs + ctr[0], ys + ctr[1], -sh * xs + ch * zs + ctr[2]], 1)   # (data_utils.py:44-70)
        recs.append({"points": pts, "seg": is_fg.astype(np.int64), "box2d": np.array([px[b] - 40, py[b] - 25, px[b] + 40, py[b] + 25]),
                     "P": P.copy(), "box3d": corners, "heading": float(heading[b]), "size": size[b].copy(),
                     "frustum_angle": float(fangle), "type": "Car"})
    return recs


def fill_state_dict(state_dict, seed=7):
    """Overwrite every entry of a torch state_dict in place from the counter hash.

    conv / head weights: zero-mean, std = sqrt(2 / fan_in) (kaiming fan_in scale,
    reference models/det_base.py:59,190); BN weight U(0.5,1.5), bias U(-0.2,0.2),
    running_mean 0, running_var 1, num_batches_tracked 0; head bias U(-0.1,0.1).
    Keys and shapes come from the module, so the same call fills the reference model
    and this package's model identically.
    """
    import torch

    for name, t in state_dict.items():
        sid = stream_id(name)
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue
        if name.endswith("running_mean"):
            t.zero_()
            continue
        if name.endswith("running_var"):
            t.fill_(1.0)
            continue
        if t.dim() >= 2:  # conv weights (Cout, Cin, k[, 1]) or ConvTranspose1d (Cin, Cout, k)
            fan_in = int(np.prod(shape[1:]))
            v = normalish(seed, sid, shape) * np.sqrt(2.0 / fan_in)
        elif name.endswith(".1.weight"):
            v = 0.5 + uniform01(seed, sid, shape)
        elif name.endswith(".1.bias"):
            v = -0.2 + 0.4 * uniform01(seed, sid, shape)
        else:  # head biases
            v = -0.1 + 0.2 * uniform01(seed, sid, shape)
        t.copy_(torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype))
    return state_dict


def to_torch(batch, device="cpu"):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in batch.items()}
