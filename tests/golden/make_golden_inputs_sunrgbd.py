#!/usr/bin/env python
"""Golden fixture for the SUN-RGBD input construction: runs the reference's OWN
datasets/provider_sample_sunrgbd.py::ProviderDataset.__getitem__ (imported read-only from /root/reference, CPU) on a small
synthetic pickle written to a temp dir, with numpy's RNG entry points wrapped so that every random draw the reference makes
(resample choice, flip coin, depth-shift normal, height-shift uniform) is recorded next to its outputs.

The fixture (tests/golden/inputs_sunrgbd_b6.npz) holds the synthetic raw records, the recorded draws and the reference's output
tensors; nothing of the reference's source is copied.  Runs only in the build container.

Usage:  python tests/golden/make_golden_inputs_sunrgbd.py
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

NPOINT = 1024
K0 = np.array([[529.5, 0.0, 365.0], [0.0, 529.5, 265.0], [0.0, 0.0, 1.0]])        # SUN-RGBD-like intrinsics (Kinect v2 scale)
CLASSES = ["bathtub", "bed", "bookshelf", "chair", "desk", "dresser", "night_stand", "sofa", "table", "toilet"]
MEAN = {"bathtub": [0.765840, 1.398258, 0.472728], "bed": [2.114256, 1.620300, 0.927272],
        "bookshelf": [0.404671, 1.071108, 1.688889], "chair": [0.591958, 0.552978, 0.827272],
        "desk": [0.695190, 1.346299, 0.736364], "dresser": [0.528526, 1.002642, 1.172878]}


def rot_x(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def synth_records(seed=20260927):
    """Six frustum records in the layout sunrgbd/prepare_data.py pickles (upright camera coordinates, NOT centre view)."""
    rng = np.random.RandomState(seed)
    counts = [300, 1024, 1500, 2600, 4000, 1023]           # < N (resample WITH replacement), == N, > N (without)
    recs = dict(id=[], box2d=[], box3d=[], inp=[], label=[], type=[], heading=[], size=[], fangle=[], K=[], R=[])
    for i, n in enumerate(counts):
        cls = CLASSES[i]
        depth = rng.uniform(1.5, 6.5)
        ang = rng.uniform(-0.5, 0.5)
        cx3, cz3 = depth * np.sin(ang), depth * np.cos(ang)
        cy3 = rng.uniform(-0.3, 0.6)
        l, w, h = [m * rng.uniform(0.9, 1.1) for m in MEAN[cls]]
        ry = rng.uniform(-np.pi, np.pi)
        c, s = np.cos(ry), np.sin(ry)
        xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
        yc = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
        zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
        corners = np.stack([c * xc + s * zc + cx3, yc + cy3, -s * xc + c * zc + cz3], 1)
        Rt = rot_x(rng.uniform(-0.25, 0.25))               # camera tilt
        K = K0.copy()
        K[0, 2] += rng.uniform(-5, 5)
        K[1, 2] += rng.uniform(-5, 5)
        # 2-D box around the image position of the object centre: invert (X,-Z,Y) <- Rtilt <- (x,z,-y) <- camera
        up = np.array([cx3, cz3, -cy3])                    # pts_3d = (X, -Z, Y)  =>  upright = (X, Y=z3, Z=-y3)
        d = Rt.T @ up                                      # depth frame (x, z, -y)
        xcam, ycam, zcam = d[0], -d[2], d[1]
        u0, v0 = K[0, 0] * xcam / zcam + K[0, 2], K[1, 1] * ycam / zcam + K[1, 2]
        bw, bh = rng.uniform(60, 200), rng.uniform(50, 160)
        box2d = np.array([u0 - bw / 2, v0 - bh / 2, u0 + bw / 2, v0 + bh / 2])
        nfg = int(0.5 * n)
        fg = np.stack([rng.uniform(-l / 2, l / 2, nfg), rng.uniform(-h / 2, h / 2, nfg), rng.uniform(-w / 2, w / 2, nfg)], 1)
        fg = np.stack([c * fg[:, 0] + s * fg[:, 2] + cx3, fg[:, 1] + cy3, -s * fg[:, 0] + c * fg[:, 2] + cz3], 1)
        dd = rng.uniform(0.4, 7.8, n - nfg)
        lat = rng.uniform(-0.12, 0.12, n - nfg)
        bg = np.stack([dd * np.sin(ang + lat), rng.uniform(-1.0, 1.2, n - nfg), dd * np.cos(ang + lat)], 1)
        pts = np.concatenate([fg, bg], 0)
        rgb = rng.uniform(0, 1, (n, 3))
        perm = rng.permutation(n)
        pts6 = np.concatenate([pts, rgb], 1)[perm].astype(np.float32)
        seg = np.concatenate([np.ones(nfg), np.zeros(n - nfg)])[perm]
        recs["id"].append(i); recs["box2d"].append(box2d); recs["box3d"].append(corners); recs["inp"].append(pts6)
        recs["label"].append(seg); recs["type"].append(cls); recs["heading"].append(ry)
        recs["size"].append(np.array([l, w, h])); recs["fangle"].append(-1.0 * np.arctan2(cz3, cx3))
        recs["K"].append(K); recs["R"].append(Rt)
    return recs


class DrawLog:
    """Wraps the numpy RNG entry points provider_sample_sunrgbd.py uses and records what they return, in call order."""

    def __init__(self):
        self.choice, self.uniform, self.normal = [], [], []
        self._c, self._r, self._n = np.random.choice, np.random.random, np.random.randn

    def __enter__(self):
        def choice(a, size=None, replace=True, p=None):
            out = self._c(a, size, replace, p)
            self.choice.append(np.asarray(out).copy())
            return out

        def random(*a):
            out = self._r(*a)
            self.uniform.append(float(out))
            return out

        def randn(*a):
            out = self._n(*a)
            self.normal.append(float(out))
            return out
        np.random.choice, np.random.random, np.random.randn = choice, random, randn
        return self

    def __exit__(self, *e):
        np.random.choice, np.random.random, np.random.randn = self._c, self._r, self._n


def main():
    import yaml
    _orig = yaml.load
    yaml.load = lambda s, Loader=None: _orig(s, Loader=Loader or yaml.FullLoader)   # configs/config.py:228 has no Loader
    sys.path.insert(0, REF)
    from configs.config import cfg, merge_cfg_from_file
    merge_cfg_from_file(os.path.join(REF, "cfgs", "det_sample_sunrgbd.yaml"))
    cfg.immutable(False)
    from datasets.provider_sample_sunrgbd import ProviderDataset, collate_fn
    recs = synth_records()
    with tempfile.TemporaryDirectory() as td:
        # (the reference's constructor ignores overwritten_data_path for the labelled splits and reads
        # <DATA_ROOT>/sunrgbd_train_aug5x.pickle, provider_sample_sunrgbd.py:79-84)
        cfg.DATA.DATA_ROOT = td
        path = os.path.join(td, "sunrgbd_train_aug5x.pickle")
        with open(path, "wb") as fp:
            pickle.dump({"id": recs["id"], "box2d": recs["box2d"], "box3d": recs["box3d"], "type": recs["type"],
                         "frustum_angle": recs["fangle"], "calib_K": recs["K"], "calib_R": recs["R"], "input": recs["inp"],
                         "label": recs["label"], "box3d_heading": recs["heading"], "box3d_size": recs["size"]}, fp)
        ds = ProviderDataset(NPOINT, split="train", random_flip=True, random_shift=True, one_hot=True,
                             overwritten_data_path=path)
        np.random.seed(777)
        items = []
        with DrawLog() as log:
            for i in range(len(ds)):
                items.append(ds[i])
        batch = collate_fn(items)
    B = len(items)
    # per sample: choice, random() (flip coin), randn() (depth shift), random() (height shift)
    assert len(log.choice) == B and len(log.uniform) == 2 * B and len(log.normal) == B
    out = {"meta_npoint": np.int64(NPOINT), "meta_strides": np.asarray(cfg.DATA.STRIDE, dtype=np.float64),
           "meta_max_depth": np.float64(cfg.DATA.MAX_DEPTH), "meta_numpy": np.bytes_(np.__version__.encode()),
           "raw_counts": np.asarray([len(p) for p in recs["inp"]], dtype=np.int64),
           "raw_points": np.concatenate(recs["inp"], 0), "raw_seg": np.concatenate(recs["label"], 0).astype(np.int64),
           "box2d": np.stack(recs["box2d"]), "K": np.stack(recs["K"]), "Rtilt": np.stack(recs["R"]),
           "box3d_corners": np.stack(recs["box3d"]), "heading": np.asarray(recs["heading"]),
           "size": np.stack(recs["size"]), "frustum_angle": np.asarray(recs["fangle"]),
           "types": np.array(recs["type"]),
           "draw_choice": np.stack(log.choice).astype(np.int32), "draw_coin": np.asarray(log.uniform[0::2]),
           "draw_normal": np.asarray(log.normal), "draw_hshift": np.asarray(log.uniform[1::2])}
    for k, v in batch.items():
        out["ref_" + k] = v.numpy()
    dst = os.path.join(HERE, "inputs_sunrgbd_b6.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: tuple(v.shape) for k, v in batch.items()})
    print("cls_label positives per sample", (batch["cls_label"] == 1).sum(1).tolist(),
          "ignore", (batch["cls_label"] == -1).sum(1).tolist())


if __name__ == "__main__":
    main()
