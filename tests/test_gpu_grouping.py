"""-m gpu: HIP sliding-frustum grouping (C-ABI fcn_query_depth_point_f32) vs the oracle: bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import grouping
from helpers import load_golden, golden_inputs, NSAMPLE, adversarial_grouping_cases

pytestmark = pytest.mark.gpu


def _gpu(dis, ns, xyz1, xyz2):
    from frustum_convnet_amd.query_depth_point import QueryDepthPoint
    op = QueryDepthPoint(dis, ns)
    idx, cnt = op(torch.from_numpy(np.ascontiguousarray(xyz1)).cuda(), torch.from_numpy(np.ascontiguousarray(xyz2)).cuda())
    assert idx.dtype == torch.int64 and cnt.dtype == torch.int32
    return idx.cpu().numpy(), cnt.cpu().numpy()


def test_testpy_scenario():
    g = load_golden("qdp_testpy")
    idx, cnt = _gpu(0.2, 4, g["xyz1"], g["xyz2"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(cnt, g["cnt"])


@pytest.mark.parametrize("case", ["car_b4_n512", "car_b4_n512_uniform", "refine_b4_n512"])
def test_golden_full(case):
    g = load_golden(case)
    d = golden_inputs(g)
    for s in range(4):
        idx, cnt = _gpu(float(g["meta_strides"][s]), NSAMPLE[s], d["point_cloud"], d["center_ref%d" % (s + 1)])
        assert np.array_equal(cnt, g["cnt%d" % (s + 1)])
        assert np.array_equal(idx, g["idx%d" % (s + 1)].astype(np.int64))


@pytest.mark.parametrize("case", ["car_b32_n1024", "people_b2_n512"])
def test_golden_sha(case):
    g = load_golden(case)
    d = golden_inputs(g)
    for s in range(4):
        idx, cnt = _gpu(float(g["meta_strides"][s]), NSAMPLE[s], d["point_cloud"], d["center_ref%d" % (s + 1)])
        assert np.array_equal(cnt, g["cnt%d" % (s + 1)])
        assert hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest() == str(g["idx%d_sha" % (s + 1)])


@pytest.mark.parametrize("B,N,M,ns,dis", [(1, 1, 1, 1, 0.5), (2, 63, 5, 7, 0.3), (2, 65, 17, 64, 0.3),
                                          (3, 1000, 33, 200, 0.05), (1, 20000, 9, 16, 0.01), (2, 300, 70, 1, 1.0)])
def test_ragged_shapes(B, N, M, ns, dis):
    rng = np.random.RandomState(N + M)
    xyz1 = rng.rand(B, 3, N).astype(np.float32)
    xyz2 = rng.rand(B, 3, M).astype(np.float32)
    xyz2[:, 2, 0] = 5.0                        # an empty window
    e_idx, e_cnt = grouping.query_depth_point_c(dis, ns, xyz1, xyz2)
    idx, cnt = _gpu(dis, ns, xyz1, xyz2)
    assert np.array_equal(cnt, e_cnt) and np.array_equal(idx, e_idx)
    assert cnt[:, 0].max() == 0 and not idx[:, 0].any()


def test_boundary_is_strict_and_fp32():
    xyz1 = np.zeros((1, 3, 8), dtype=np.float32)
    xyz1[0, 2] = [5.0, 1.0, 1.25, 0.75, 1.0, 1.1, 0.9, 1.0]
    xyz2 = np.zeros((1, 3, 3), dtype=np.float32)
    xyz2[0, 2] = [1.0, 3.0, 5.25]
    idx, cnt = _gpu(0.25, 3, xyz1, xyz2)
    assert cnt.tolist() == [[3, 0, 0]] and idx[0, 0].tolist() == [1, 4, 5]
    idx, cnt = _gpu(0.2500001, 8, xyz1, xyz2)
    assert cnt.tolist() == [[7, 0, 1]] and idx[0, 0].tolist() == [1, 2, 3, 4, 5, 6, 7, 1]


def test_kernel_native_layout_matches():
    from frustum_convnet_amd.query_depth_point import query_depth_point, query_depth_point_bn3
    g = load_golden("car_b4_n512")
    d = golden_inputs(g)
    pc = torch.from_numpy(d["point_cloud"]).cuda()
    ref = torch.from_numpy(d["center_ref2"]).cuda()
    a = query_depth_point(0.5, 64, pc, ref)
    b = query_depth_point_bn3(0.5, 64, pc.permute(0, 2, 1).contiguous(), ref.permute(0, 2, 1).contiguous())
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_double_inputs_are_narrowed_like_the_reference_kernel():
    """The reference dispatches double (query_depth_point_cuda_kernel.cu:77) and narrows both depths to float inside
    (.cu:40,48): the module must return what the fp32 op returns on the narrowed inputs."""
    from frustum_convnet_amd.query_depth_point import QueryDepthPoint
    rng = np.random.RandomState(3)
    xyz1 = torch.from_numpy(rng.uniform(-1, 1, (2, 3, 200))).cuda()            # float64
    xyz2 = torch.from_numpy(rng.uniform(-1, 1, (2, 3, 17))).cuda()
    op = QueryDepthPoint(0.2, 8)
    i64, c64 = op(xyz1, xyz2)
    i32, c32 = op(xyz1.float(), xyz2.float())
    assert torch.equal(i64, i32) and torch.equal(c64, c32)
    e_idx, e_cnt = grouping.query_depth_point_numpy(0.2, 8, xyz1.float().cpu().numpy(), xyz2.float().cpu().numpy())
    assert np.array_equal(i64.cpu().numpy(), e_idx) and np.array_equal(c64.cpu().numpy(), e_cnt)


def test_adversarial_inputs():
    """ NaN / +-inf depths, dis_z in {0, < 0, inf}, nsample > n, >= 65 equal-depth hits straddling the nsample
    cut and the 64-lane chunks, N just above the LDS staging limits: the API kernel (int64 idx + cnt) against both oracle
    restatements, and the fused front's counts / entry lists against the compacted oracle index."""
    import entry_ref
    from frustum_convnet_amd import pointnet_fused as pf
    for name, dis, ns, xyz1, xyz2 in adversarial_grouping_cases():
        with np.errstate(invalid="ignore"):
            e_idx, e_cnt = grouping.query_depth_point_c(dis, ns, xyz1, xyz2)
            n_idx, n_cnt = grouping.query_depth_point_numpy(dis, ns, xyz1, xyz2)
        assert np.array_equal(e_idx, n_idx) and np.array_equal(e_cnt, n_cnt), name
        idx, cnt = _gpu(dis, ns, xyz1, xyz2)
        assert np.array_equal(cnt, e_cnt) and np.array_equal(idx, e_idx), name
        # the model's fused front on the same inputs (it never builds idx: compare the compacted form)
        if ns > 1024 or not np.isfinite(xyz1).all() or not np.isfinite(xyz2).all():
            continue        # (entry rows of non-finite points are NaN != NaN; K is bounded by the descriptor)
        B, M = xyz2.shape[0], xyz2.shape[2]
        pc, ref = torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda()
        C = (64, 64, 128)
        g = torch.Generator().manual_seed(5)
        plist = []
        for i, (co, ci) in enumerate(((C[0], 3), (C[1], C[0]), (C[2], C[1]))):
            plist += [(torch.randn(co, ci, generator=g) * 0.1).cuda(), torch.ones(co).cuda(), torch.zeros(co).cuda()]
        bufs = ([torch.zeros(c).cuda() for c in C], [torch.ones(c).cuda() for c in C],
                [torch.zeros((), dtype=torch.int64).cuda() for c in C])
        pool = pf.WorkspacePool()
        h = pf._acquire(pool, (float(dis), ns, True, 1e-5, 0.1, False, True), pc, ref, None, bufs, plist, False)
        pf.group_compact([h], pc)
        torch.cuda.synchronize()
        c = entry_ref.compact(torch.from_numpy(e_idx), torch.from_numpy(e_cnt), torch.from_numpy(xyz1), torch.from_numpy(xyz2), ns)
        ws = h["ws"]
        assert torch.equal(ws.cnt.cpu(), torch.from_numpy(e_cnt)), name
        assert torch.equal(ws.woff.cpu(), c["woff"]), name
        for b in range(B):
            n = int(c["nent"][b])
            assert torch.equal(ws.ent.cpu()[b, :n], c["ent"][b, :n]) and torch.equal(ws.ewin.cpu()[b, :n], c["ewin"][b, :n]), (name, b)


def test_multi_scale_launch_equals_the_per_scale_operator():
    """fcn_query_depth_point_multi_f32: all scales of a batch in one launch -- idx / cnt bit for bit those of one call of the operator
    per scale (car scales on the golden batch, a ragged one with nsample > n and an empty scale)."""
    from frustum_convnet_amd.query_depth_point import query_depth_point, query_depth_point_multi
    d = golden_inputs(load_golden("car_b4_n512"))
    pc = torch.from_numpy(d["point_cloud"][:, :3].copy()).cuda().contiguous()
    refs = [torch.from_numpy(d["center_ref%d" % i].copy()).cuda().contiguous() for i in (1, 2, 3, 4)]
    dz, ks = [0.25, 0.5, 1.0, 2.0], [32, 64, 64, 128]
    got = query_depth_point_multi(dz, ks, pc, refs)
    for s in range(4):
        idx, cnt = query_depth_point(dz[s], ks[s], pc, refs[s])
        assert torch.equal(got[s][0], idx) and torch.equal(got[s][1], cnt)
    again = query_depth_point_multi(dz, ks, pc, refs, out=got)           # in place
    assert again is got
    gen = torch.Generator().manual_seed(5)
    pc2 = (torch.rand((3, 3, 70), generator=gen) * 4).cuda()
    refs2 = [(torch.rand((3, 3, m), generator=gen) * 4).cuda() for m in (17, 1, 33)]
    dz2, ks2 = [0.3, 5.0, 0.01], [100, 7, 16]
    got2 = query_depth_point_multi(dz2, ks2, pc2, refs2)
    for s in range(3):
        idx, cnt = query_depth_point(dz2[s], ks2[s], pc2, refs2[s])
        assert torch.equal(got2[s][0], idx) and torch.equal(got2[s][1], cnt)
