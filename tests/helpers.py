"""Shared test helpers: load golden fixtures, rebuild their inputs from seeds, build state_dicts."""
import os

import numpy as np
import torch

from frustum_convnet_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NSAMPLE = (32, 64, 64, 128)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_inputs(g):
    zr = g["meta_z_range"]
    z_range = None if np.isnan(zr[0]) else (float(zr[0]), float(zr[1]))
    if len(g["meta_strides"]) == 5:          # the SUN-RGBD fixture (tests/golden/make_golden.py sunrgbd): 8 m, 10 classes
        from oracle.det_ref import MEAN_SIZE_SUNRGBD
        return synth.make_batch(int(g["meta_batch"]), int(g["meta_npoint"]), strides=tuple(g["meta_strides"]),
                                max_depth=8.0, seed=int(g["meta_seed"]), variant=str(g["meta_variant"]),
                                tilt=tuple(g["meta_tilt"]), z_range=z_range, num_classes=10,
                                mean_sizes=MEAN_SIZE_SUNRGBD)
    return synth.make_batch(int(g["meta_batch"]), int(g["meta_npoint"]), strides=tuple(g["meta_strides"]),
                            seed=int(g["meta_seed"]), variant=str(g["meta_variant"]),
                            tilt=tuple(g["meta_tilt"]), z_range=z_range)


def golden_state_dict(g, seed=7, dtype=torch.float32):
    """state_dict with the reference's key names/shapes (recorded in the fixture), hash-filled."""
    sd = {}
    for k, s in zip(g["state_keys"], g["state_shapes"]):
        shape = tuple(int(x) for x in str(s).strip("()").split(",") if x.strip())
        sd[str(k)] = torch.zeros(shape, dtype=torch.int64 if str(k).endswith("num_batches_tracked") else dtype)
    synth.fill_state_dict(sd, seed=seed)
    return sd
