#!/usr/bin/env python
"""Golden fixture for the on-device input construction: runs the reference's OWN
datasets/provider_sample.py::ProviderDataset.__getitem__ (imported read-only from /root/reference, CPU) on a small
synthetic pickle written to a temp dir, with numpy's RNG entry points wrapped so that every random draw the
reference makes (resample choice, flip coin, shift normal) is recorded next to its outputs.

The fixture (tests/golden/inputs_kitti_b6.npz) holds the synthetic raw records, the recorded draws and the
reference's output tensors; nothing of the reference's source is copied.  Runs only in the build container.

Usage:  python tests/golden/make_golden_inputs.py
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

NPOINT = 512
P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
               [0.0, 721.5377, 172.854, 0.2163791],
               [0.0, 0.0, 1.0, 0.002745884]], dtype=np.float64)     # a KITTI P2 (public calibration values)


def synth_records(seed=20260926):
    """Six frustum records in the layout kitti/prepare_data.py pickles (rect camera coordinates, NOT centre view)."""
    rng = np.random.RandomState(seed)
    counts = [180, 512, 700, 1500, 3000, 511]              # < N (resample WITH replacement), == N, > N
    recs = dict(id=[], box2d=[], box3d=[], inp=[], label=[], type=[], heading=[], size=[], fangle=[], gtbox2d=[], calib=[])
    for i, n in enumerate(counts):
        depth = rng.uniform(8.0, 55.0)
        ang = rng.uniform(-0.6, 0.6)                        # direction of the frustum axis w.r.t. +z
        cx3, cz3 = depth * np.sin(ang), depth * np.cos(ang)
        cy3 = rng.uniform(0.6, 1.2)
        l, w, h = 3.88 * rng.uniform(0.9, 1.1), 1.63 * rng.uniform(0.9, 1.1), 1.53 * rng.uniform(0.9, 1.1)
        ry = rng.uniform(-np.pi, np.pi)
        c, s = np.cos(ry), np.sin(ry)
        xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
        yc = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
        zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
        corners = np.stack([c * xc + s * zc + cx3, yc + cy3, -s * xc + c * zc + cz3], 1)
        # 2-D box: projection of the centre +- a few pixels (only its centre is used by generate_ref)
        u0 = P2[0, 0] * cx3 / cz3 + P2[0, 2] + P2[0, 3] / cz3
        v0 = P2[1, 1] * cy3 / cz3 + P2[1, 2] + P2[1, 3] / cz3
        bw, bh = rng.uniform(40, 160), rng.uniform(30, 90)
        box2d = np.array([u0 - bw / 2, v0 - bh / 2, u0 + bw / 2, v0 + bh / 2])
        # points: 55 % on the object, the rest spread along the frustum axis; float32 xyz + intensity
        nfg = int(0.55 * n)
        fg = np.stack([rng.uniform(-l / 2, l / 2, nfg), rng.uniform(-h / 2, h / 2, nfg), rng.uniform(-w / 2, w / 2, nfg)], 1)
        fg = np.stack([c * fg[:, 0] + s * fg[:, 2] + cx3, fg[:, 1] + cy3, -s * fg[:, 0] + c * fg[:, 2] + cz3], 1)
        d = rng.uniform(2.0, 68.0, n - nfg)
        lat = rng.uniform(-0.06, 0.06, n - nfg)
        bg = np.stack([d * np.sin(ang + lat), rng.uniform(-1.0, 1.8, n - nfg), d * np.cos(ang + lat)], 1)
        pts = np.concatenate([fg, bg], 0)
        inten = rng.uniform(0, 1, (n, 1))
        perm = rng.permutation(n)
        pts4 = np.concatenate([pts, inten], 1)[perm].astype(np.float32)
        seg = np.concatenate([np.ones(nfg), np.zeros(n - nfg)])[perm]
        recs["id"].append(i); recs["box2d"].append(box2d); recs["box3d"].append(corners); recs["inp"].append(pts4)
        recs["label"].append(seg); recs["type"].append("Car"); recs["heading"].append(ry)
        recs["size"].append(np.array([l, w, h])); recs["fangle"].append(-1.0 * np.arctan2(cz3, cx3))
        recs["gtbox2d"].append(box2d.copy()); recs["calib"].append({"P2": P2.reshape(-1).copy()})
    return recs


class DrawLog:
    """Wraps the three numpy RNG entry points provider_sample.py uses and records what they return."""

    def __init__(self):
        self.choice, self.coin, self.normal = [], [], []
        self._c, self._r, self._n = np.random.choice, np.random.random, np.random.randn

    def __enter__(self):
        def choice(a, size=None, replace=True, p=None):
            out = self._c(a, size, replace, p)
            self.choice.append(np.asarray(out).copy())
            return out

        def random(*a):
            out = self._r(*a)
            self.coin.append(float(out))
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
    merge_cfg_from_file(os.path.join(REF, "cfgs", "det_sample.yaml"))
    cfg.immutable(False)
    from datasets.provider_sample import ProviderDataset, collate_fn
    recs = synth_records()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "synthetic_frustums.pickle")
        with open(path, "wb") as fp:
            for k in ("id", "box2d", "box3d", "inp", "label", "type", "heading", "size", "fangle", "gtbox2d", "calib"):
                pickle.dump(recs[k], fp)
        ds = ProviderDataset(NPOINT, split="train", random_flip=True, random_shift=True, one_hot=True,
                             overwritten_data_path=path)
        np.random.seed(4242)
        items = []
        with DrawLog() as log:
            for i in range(len(ds)):
                items.append(ds[i])
        batch = collate_fn(items)
    B = len(items)
    assert len(log.choice) == B and len(log.coin) == B and len(log.normal) == B
    out = {"meta_npoint": np.int64(NPOINT), "meta_strides": np.asarray(cfg.DATA.STRIDE, dtype=np.float64),
           "meta_max_depth": np.float64(cfg.DATA.MAX_DEPTH), "meta_numpy": np.bytes_(np.__version__.encode()),
           "raw_counts": np.asarray([len(p) for p in recs["inp"]], dtype=np.int64),
           "raw_points": np.concatenate(recs["inp"], 0), "raw_seg": np.concatenate(recs["label"], 0).astype(np.int64),
           "box2d": np.stack(recs["box2d"]), "P": np.stack([c["P2"].reshape(3, 4) for c in recs["calib"]]),
           "box3d_corners": np.stack(recs["box3d"]), "heading": np.asarray(recs["heading"]),
           "size": np.stack(recs["size"]), "frustum_angle": np.asarray(recs["fangle"]),
           "draw_choice": np.stack(log.choice).astype(np.int32), "draw_coin": np.asarray(log.coin),
           "draw_normal": np.asarray(log.normal)}
    for k, v in batch.items():
        out["ref_" + k] = v.numpy()
    dst = os.path.join(HERE, "inputs_kitti_b6.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: tuple(v.shape) for k, v in batch.items()})
    print("cls_label positives per sample", (batch["cls_label"] == 1).sum(1).tolist(),
          "ignored", (batch["cls_label"] == -1).sum(1).tolist(), "flips", [c > 0.5 for c in log.coin])


if __name__ == "__main__":
    main()
