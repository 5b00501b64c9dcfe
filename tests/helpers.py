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


def adversarial_grouping_cases():
    """(name, dis_z, nsample, xyz1 (B,3,N), xyz2 (B,3,M)) -- inputs the reference kernel's predicate `fabsf(z2 - z1) < dis_z`
    (query_depth_point_cuda_kernel.cu:40-64) treats in ways a restructured scan (ballots over 64-lane chunks, prefix popcounts,
    early exit at nsample) could get wrong: non-finite depths, a degenerate threshold, nsample above the point count, long runs
    of equal depths straddling the nsample cut and the 64-lane chunk boundaries, N just above the LDS staging limits."""
    import numpy as np
    rng = np.random.RandomState(2024)
    out = []

    def mk(zp, zc):
        zp, zc = np.atleast_2d(np.asarray(zp, np.float32)), np.atleast_2d(np.asarray(zc, np.float32))
        a = rng.rand(zp.shape[0], 3, zp.shape[1]).astype(np.float32)
        b = rng.rand(zc.shape[0], 3, zc.shape[1]).astype(np.float32)
        a[:, 2, :] = zp
        b[:, 2, :] = zc
        return a, b
    # NaN / +-inf depths among the points and among the window centres
    zp = rng.rand(2, 150).astype(np.float32)
    zp[0, [0, 7, 64, 65, 149]] = np.nan
    zp[1, [3, 63, 64, 128]] = [np.inf, -np.inf, np.inf, np.nan]
    zc = rng.rand(2, 9).astype(np.float32)
    zc[0, 2], zc[0, 5], zc[1, 1], zc[1, 4] = np.nan, np.inf, -np.inf, np.nan
    for ns in (4, 200):
        out.append(("nonfinite_ns%d" % ns, 0.3, ns) + mk(zp, zc))
    out.append(("nonfinite_dis_inf", float("inf"), 16) + mk(zp, zc))
    # dis_z = 0 (no hit possible: strict <) and dis_z < 0
    zp = np.tile(np.linspace(0, 1, 70, dtype=np.float32), (1, 1))
    for dis in (0.0, -1.0):
        out.append(("dis_%g" % dis, dis, 8) + mk(zp, zp[:, :5]))
    # nsample larger than the number of points
    out.append(("nsample_gt_n", 0.5, 40) + mk(rng.rand(2, 13), rng.rand(2, 6)))
    # >= 65 equal depths: the run crosses 64-lane chunk boundaries and the nsample cut lands inside / at the edges of it
    zp = rng.rand(1, 400).astype(np.float32) + 10.0
    zp[0, 30:30 + 130] = 1.0                                   # 130 equal-depth hits: points 30..159
    zp[0, 250:260] = 1.0
    zc = np.array([[1.0, 1.0000001, 0.5, 10.5]], np.float32)
    for ns in (1, 34, 63, 64, 65, 100, 129, 130, 131, 140, 141):
        out.append(("equal_run_ns%d" % ns, 0.25, ns) + mk(zp, zc))
    # every point hits (dis_z huge) with N not a multiple of 64
    out.append(("all_hit", 1e9, 77) + mk(rng.rand(2, 191), rng.rand(2, 3)))
    # N just above the staging limits of the API kernel (16384 points) and of the fused front (8192)
    for n in (8193, 16385, 16449):
        zp = rng.rand(1, n).astype(np.float32)
        zc = np.array([[zp[0, -1], zp[0, n // 2], 0.5, 2.0]], np.float32)      # windows whose hits include the LAST point
        out.append(("n_%d" % n, 2e-4, 32) + mk(zp, zc))
    return out
