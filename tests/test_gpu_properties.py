"""-m gpu: size-independent properties at BASELINE.json's full size (KITTI-car, B=32, N=1024, 4 strides), where the CPU
oracle is too slow to be the checker for every element."""
import numpy as np
import pytest
import torch

from frustum_convnet_amd import synth

pytestmark = pytest.mark.gpu
B, N = 32, 1024


def _full_batch(seed=4321):
    return synth.to_torch(synth.make_batch(B, N, seed=seed, variant="car", tilt=(0.01, 0.05)), "cuda")


def _model(seed=7):
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    synth.fill_state_dict(m.state_dict(), seed=seed)
    return m.cuda()


def test_grouping_structure_full_size():
    """Every window: the first cnt slots ascend strictly (points are taken in index order), lie inside the window
    (|z - zc| < dis_z, strict, fp32), the remaining slots repeat the first hit, cnt = min(hits, nsample) against a
    brute-force fp32 count, empty windows are all-zero."""
    from frustum_convnet_amd.query_depth_point import query_depth_point
    data = _full_batch()
    pc = data["point_cloud"][:, :3].contiguous()
    z = pc[:, 2, :]
    for s, (dist, K) in enumerate(((0.25, 32), (0.5, 64), (1.0, 64), (2.0, 128))):
        ref = data["center_ref%d" % (s + 1)].contiguous()
        idx, cnt = query_depth_point(dist, K, pc, ref)
        zc = ref[:, 2, :]
        inside = (zc[:, :, None] - z[:, None, :]).abs() < np.float32(dist)           # (B, L, N), fp32 compare
        hits = inside.sum(-1)
        assert torch.equal(cnt.long(), hits.clamp(max=K))
        k = torch.arange(K, device="cuda").view(1, 1, K)
        live = k < cnt.unsqueeze(-1)
        zi = torch.gather(z.unsqueeze(1).expand(-1, idx.shape[1], -1), 2, idx)
        assert bool((((zi - zc.unsqueeze(-1)).abs() < np.float32(dist)) | ~live | (cnt == 0).unsqueeze(-1)).all())
        asc = (idx[:, :, 1:] > idx[:, :, :-1]) | ~live[:, :, 1:]
        assert bool(asc.all())
        pad_ok = (idx == idx[:, :, :1]) | live
        assert bool(pad_ok.all())
        assert bool((idx[cnt == 0] == 0).all())
        # the k-th live slot is the k-th inside point: rank of the chosen point among the inside points
        rank = torch.cumsum(inside.long(), -1) - 1
        ri = torch.gather(rank, 2, idx)
        assert bool(((ri == k) | ~live).all())


def test_full_step_is_bitwise_reproducible():
    """Two runs of the same full-size step give identical logits and identical gradients, bit for bit (no atomics on
    activations or gradients; BN sums are fp64 atomics of per-tile fp32 partials)."""
    data = _full_batch()
    outs = []
    for _ in range(2):
        m = _model()
        m.train()
        lo, _ = m(data)
        lo["total_loss"].backward()
        cls, reg = m.last_logits
        outs.append((cls.detach().clone(), reg.detach().clone(),
                     torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])


def test_eval_outputs_are_independent_of_the_rest_of_the_batch():
    """Inference uses running statistics, so a frustum's outputs do not depend on its batch mates: the full batch against
    the same frustums run alone / in a different batch composition (different tile shapes and summation order of the
    GEMMs: 1e-4, the logits' parity tolerance)."""
    data = _full_batch()
    m = _model()
    m.eval()
    keys = ("point_cloud", "one_hot", "center_ref1", "center_ref2", "center_ref3", "center_ref4")
    with torch.no_grad():
        m({k: data[k] for k in keys})
        cls_all, reg_all = [t.clone() for t in m.last_logits]
        for sel in ([0], [5, 17], list(range(31, 15, -1))):
            m({k: data[k][sel].contiguous() for k in keys})
            cls, reg = m.last_logits
            assert float((cls - cls_all[sel]).abs().max()) < 1e-4
            assert float((reg - reg_all[sel]).abs().max()) < 1e-4


def test_train_statistics_are_the_only_coupling_between_frustums():
    """Training-mode BatchNorm couples the frustums ONLY through the batch statistics: duplicating the whole batch
    (2B frustums) leaves mean / biased variance unchanged, hence the logits of the first copy (1e-4)."""
    data = _full_batch()
    half = {k: v[:8].contiguous() for k, v in data.items()}
    dup = {k: torch.cat([v, v], 0) for k, v in half.items()}
    m1, m2 = _model(), _model()
    m1.train()
    m2.train()
    with torch.no_grad():
        m1(half)
        c1, r1 = [t.clone() for t in m1.last_logits]
        m2(dup)
        c2, r2 = m2.last_logits
    assert float((c2[:8] - c1).abs().max()) < 1e-4 and float((c2[8:] - c1).abs().max()) < 1e-4
    assert float((r2[:8] - r1).abs().max()) < 1e-4 and float((r2[8:] - r1).abs().max()) < 1e-4
