#!/usr/bin/env python
"""Golden fixture for the on-device REFINE-stage input construction (refine variant; BASELINE
config 5): runs the reference's OWN datasets/provider_sample_refine.py::ProviderDataset.__getitem__ + collate_fn (imported
read-only from /root/reference, CPU) on a small synthetic pickle, with numpy's RNG entry points wrapped so that every random
draw the reference makes is recorded next to its outputs.  Per-sample L differs (the predicted boxes have different depths
extents): the fixture exercises the edge padding of collate_fn.

Stores raw records, recorded draws and the reference's collated batch; nothing of the reference's source is copied.
Usage:  python tests/golden/make_golden_inputs_refine.py
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
sys.path.insert(0, HERE)

from make_golden_inputs import DrawLog, P2  # noqa: E402

NPOINT = 512


def corners_of(center, size, ry):
    l, w, h = size
    c, s = np.cos(ry), np.sin(ry)
    xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    yc = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
    zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    return np.stack([c * xc + s * zc + center[0], yc + center[1], -s * xc + c * zc + center[2]], 1)


def synth_records(seed=20260927):
    rng = np.random.RandomState(seed)
    counts = [90, 512, 640, 1300, 300, 511]
    recs = dict(id=[], box3d=[], inp=[], label=[], type=[], heading=[], size=[], fangle=[], box2d=[], calib=[],
                pred=[], pred_size=[], pred_angle=[])
    for i, n in enumerate(counts):
        depth = rng.uniform(8.0, 45.0)
        ang = rng.uniform(-0.5, 0.5)
        ctr = np.array([depth * np.sin(ang), rng.uniform(0.6, 1.2), depth * np.cos(ang)])
        size = np.array([3.88, 1.63, 1.53]) * rng.uniform(0.85, 1.15, 3)
        ry = rng.uniform(-np.pi, np.pi)
        # the first-stage prediction: a perturbed, 1.2x enlarged box (kitti/prepare_data_refine.py:319-321)
        pctr = ctr + rng.normal(0, 0.25, 3)
        psize = size * rng.uniform(0.9, 1.1, 3) * 1.2
        pry = ry + rng.normal(0, 0.15)
        nfg = int(0.6 * n)
        fg = np.stack([rng.uniform(-0.5, 0.5, nfg) * size[0], rng.uniform(-0.5, 0.5, nfg) * size[2], rng.uniform(-0.5, 0.5, nfg) * size[1]], 1)
        c, s = np.cos(ry), np.sin(ry)
        fg = np.stack([c * fg[:, 0] + s * fg[:, 2] + ctr[0], fg[:, 1] + ctr[1], -s * fg[:, 0] + c * fg[:, 2] + ctr[2]], 1)
        bg = pctr[None, :] + rng.uniform(-1, 1, (n - nfg, 3)) * np.array([3.0, 1.0, 3.0])
        pts = np.concatenate([fg, bg], 0)
        perm = rng.permutation(n)
        pts4 = np.concatenate([pts, rng.uniform(0, 1, (n, 1))], 1)[perm].astype(np.float32)
        recs["id"].append(i); recs["box3d"].append(corners_of(ctr, size, ry)); recs["inp"].append(pts4)
        recs["label"].append(np.concatenate([np.ones(nfg), np.zeros(n - nfg)])[perm]); recs["type"].append("Car")
        recs["heading"].append(ry); recs["size"].append(size); recs["fangle"].append(-1.0 * np.arctan2(ctr[2], ctr[0]))
        recs["box2d"].append(np.array([100.0, 100.0, 200.0, 180.0])); recs["calib"].append({"P2": P2.reshape(-1).copy()})
        recs["pred"].append(corners_of(pctr, psize, pry)); recs["pred_size"].append(psize); recs["pred_angle"].append(pry)
    return recs


def main():
    import yaml
    _orig = yaml.load
    yaml.load = lambda s, Loader=None: _orig(s, Loader=Loader or yaml.FullLoader)
    sys.path.insert(0, REF)
    from configs.config import cfg, merge_cfg_from_file
    merge_cfg_from_file(os.path.join(REF, "cfgs", "refine_car.yaml"))
    cfg.immutable(False)
    from datasets.provider_sample_refine import ProviderDataset, collate_fn
    recs = synth_records()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "synthetic_refine.pickle")
        with open(path, "wb") as fp:
            for k in ("id", "box3d", "inp", "label", "type", "heading", "size", "fangle", "box2d", "calib", "pred", "pred_size",
                      "pred_angle"):
                pickle.dump(recs[k], fp)
        ds = ProviderDataset(NPOINT, split="train", random_flip=True, random_shift=True, one_hot=True,
                             overwritten_data_path=path)
        np.random.seed(777)
        items = []
        with DrawLog() as log:
            for i in range(len(ds)):
                items.append(ds[i])
        lens = [[it["center_ref%d" % s].shape[1] for s in (1, 2, 3, 4)] for it in items]
        batch = collate_fn(items)
    B = len(items)
    assert len(log.choice) == B and len(log.coin) == B and len(log.normal) == B
    out = {"meta_npoint": np.int64(NPOINT), "meta_strides": np.asarray(cfg.DATA.STRIDE, dtype=np.float64),
           "raw_counts": np.asarray([len(p) for p in recs["inp"]], dtype=np.int64), "raw_points": np.concatenate(recs["inp"], 0),
           "box3d_corners": np.stack(recs["box3d"]), "heading": np.asarray(recs["heading"]), "size": np.stack(recs["size"]),
           "pred_corners": np.stack(recs["pred"]), "pred_size": np.stack(recs["pred_size"]),
           "pred_angle": np.asarray(recs["pred_angle"]), "draw_choice": np.stack(log.choice).astype(np.int32),
           "draw_coin": np.asarray(log.coin), "draw_normal": np.asarray(log.normal), "ref_lens": np.asarray(lens, dtype=np.int64)}
    for k, v in batch.items():
        out["ref_" + k] = v.numpy() if hasattr(v, "numpy") else np.asarray(v)
    dst = os.path.join(HERE, "inputs_refine_b6.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: tuple(np.asarray(v).shape) for k, v in batch.items()})
    print("per-sample L:", lens, "positives", (batch["cls_label"] == 1).sum(1).tolist(), "flips", [c > 0.5 for c in log.coin])


if __name__ == "__main__":
    main()
