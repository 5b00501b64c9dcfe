"""Pins the grouping oracle (oracle/qdp_ref.c + oracle/grouping.py).

The reference op is CUDA-only; the single artefact that states its expected behaviour is
ops/query_depth_point/test.py:10-28 -- a CPU mask ``abs(z - z1) < 0.2`` printed next to the op's
idx for eyeballing.  We assert what that script lets a reader check: row = first `nsample`
nonzero mask positions in ascending order, short rows padded with the first hit, cnt = min(hits, nsample).
"""
import numpy as np
import pytest

from oracle import grouping
from helpers import load_golden, golden_inputs, NSAMPLE, adversarial_grouping_cases


def _expect_from_mask(mask, nsample):
    B, M, N = mask.shape
    idx = np.zeros((B, M, nsample), dtype=np.int64)
    cnt = np.zeros((B, M), dtype=np.int32)
    for b in range(B):
        for m in range(M):
            hits = np.nonzero(mask[b, m])[0][:nsample]
            cnt[b, m] = len(hits)
            if len(hits):
                idx[b, m, :] = hits[0]
                idx[b, m, :len(hits)] = hits
    return idx, cnt


def test_testpy_mask_criterion():
    g = load_golden("qdp_testpy")
    exp_idx, exp_cnt = _expect_from_mask(g["mask"].astype(bool), 4)
    for fn in (grouping.query_depth_point_c, grouping.query_depth_point_numpy):
        idx, cnt = fn(0.2, 4, g["xyz1"], g["xyz2"])
        assert idx.dtype == np.int64 and cnt.dtype == np.int32
        assert np.array_equal(idx, exp_idx)
        assert np.array_equal(cnt, exp_cnt)
    assert np.array_equal(g["idx"], exp_idx) and np.array_equal(g["cnt"], exp_cnt)
    # every query is one of the points, so every window holds at least itself
    assert (exp_cnt >= 1).all()


@pytest.mark.parametrize("case", ["car_b4_n512", "car_b4_n512_uniform", "refine_b4_n512"])
def test_matches_golden_full_idx(case):
    g = load_golden(case)
    data = golden_inputs(g)
    for s in range(4):
        idx, cnt = grouping.query_depth_point_c(float(g["meta_strides"][s]), NSAMPLE[s],
                                                data["point_cloud"], data["center_ref%d" % (s + 1)])
        assert np.array_equal(cnt, g["cnt%d" % (s + 1)])
        assert np.array_equal(idx, g["idx%d" % (s + 1)].astype(np.int64))
        i2, c2 = grouping.query_depth_point_numpy(float(g["meta_strides"][s]), NSAMPLE[s],
                                                  data["point_cloud"], data["center_ref%d" % (s + 1)])
        assert np.array_equal(idx, i2) and np.array_equal(cnt, c2)


def test_edge_cases():
    # empty window -> zeros, cnt 0; overflow window -> first nsample in index order; strict <
    xyz1 = np.zeros((1, 3, 8), dtype=np.float32)
    xyz1[0, 2] = [5.0, 1.0, 1.25, 0.75, 1.0, 1.1, 0.9, 1.0]
    xyz2 = np.zeros((1, 3, 3), dtype=np.float32)
    xyz2[0, 2] = [1.0, 3.0, 5.25]
    for fn in (grouping.query_depth_point_c, grouping.query_depth_point_numpy):
        idx, cnt = fn(0.25, 3, xyz1, xyz2)
        assert cnt.tolist() == [[3, 0, 0]]            # |5.25-5.0| = 0.25 is NOT < 0.25
        assert idx[0, 0].tolist() == [1, 4, 5]        # 1.25 and 0.75 sit exactly on the boundary
        assert idx[0, 1].tolist() == [0, 0, 0] and idx[0, 2].tolist() == [0, 0, 0]
        idx, cnt = fn(0.2500001, 8, xyz1, xyz2)
        assert cnt.tolist() == [[7, 0, 1]]
        assert idx[0, 0].tolist() == [1, 2, 3, 4, 5, 6, 7, 1]
        assert idx[0, 2].tolist() == [0] * 8
    # zero-size inputs
    idx, cnt = grouping.query_depth_point_c(1.0, 4, np.zeros((2, 3, 0), np.float32), np.zeros((2, 3, 5), np.float32))
    assert idx.shape == (2, 5, 4) and not idx.any() and not cnt.any()


def test_adversarial_inputs_both_restatements_and_the_mask_criterion_agree():
    """NaN / inf depths, dis_z <= 0, nsample > n, equal-depth runs across the nsample cut, N above the staging limits: the C
    restatement, the independent numpy formulation and test.py's mask criterion give the same idx / cnt (the GPU test
    test_gpu_grouping.py::test_adversarial_inputs holds the HIP kernels to the same arrays)."""
    for name, dis, ns, xyz1, xyz2 in adversarial_grouping_cases():
        with np.errstate(invalid="ignore"):
            mask = np.abs(xyz2[:, 2, :, None] - xyz1[:, 2, None, :]) < np.float32(dis)
            e_idx, e_cnt = _expect_from_mask(mask, ns)
            ic, cc = grouping.query_depth_point_c(dis, ns, xyz1, xyz2)
            inp, cn = grouping.query_depth_point_numpy(dis, ns, xyz1, xyz2)
        assert np.array_equal(ic, e_idx) and np.array_equal(cc, e_cnt), name
        assert np.array_equal(inp, e_idx) and np.array_equal(cn, e_cnt), name
