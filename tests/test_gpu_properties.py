"""-m gpu: size-independent properties at BASELINE.json's full sizes (B=32: KITTI-car N=1024 L=(280,140,70,35); people
N=1024 L=(700,350,175,88); refine N=512 L=(20,10,5,3); SUN-RGBD N=2048, 5 scales L=(80,40,20,10,5)), where the CPU oracle
is too slow to be the checker for every element."""
import numpy as np
import pytest
import torch

from frustum_convnet_amd import synth

pytestmark = pytest.mark.gpu
B, N = 32, 1024


# name: (strides, z_range, N, nsample per scale)
CONFIGS = {
    "car": ((0.25, 0.5, 1.0, 2.0), None, 1024, (32, 64, 64, 128)),
    "people": ((0.1, 0.2, 0.4, 0.8), None, 1024, (32, 64, 64, 128)),
    "refine": ((0.1, 0.2, 0.4, 0.8), (-1.0, 1.0), 512, (32, 64, 64, 128)),
    "sunrgbd": ((0.1, 0.2, 0.4, 0.8, 1.6), None, 2048, (128, 128, 256, 256, 256)),
}


def _full_batch(seed=4321, cfg_name="car"):
    strides, z_range, npoint, _ = CONFIGS[cfg_name]
    if cfg_name == "sunrgbd":
        from frustum_convnet_amd.dataset_info import SUNRGBDCategory
        d = synth.make_batch(B, npoint, strides=strides, max_depth=8.0, seed=seed, variant="car", tilt=(0.01, 0.05),
                             num_classes=10, mean_sizes=SUNRGBDCategory.MEAN_SIZE_ARRAY)
    else:
        d = synth.make_batch(B, npoint, strides=strides, seed=seed, variant="car", tilt=(0.01, 0.05), z_range=z_range)
    return synth.to_torch(d, "cuda")


def _model(seed=7, cfg_name="car"):
    from frustum_convnet_amd.config import cfg, reset_cfg
    from frustum_convnet_amd import det_base, det_base_sunrgbd
    reset_cfg()
    cfg.DATA.HEIGHT_HALF = CONFIGS[cfg_name][0]
    cfg.DATA.STRIDE = CONFIGS[cfg_name][0]
    if cfg_name == "sunrgbd":
        cfg.DATA.DATASET_NAME = "SUNRGBD"
        m = det_base_sunrgbd.PointNetDet(3, num_vec=10, num_classes=2)
    else:
        m = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    synth.fill_state_dict(m.state_dict(), seed=seed)
    return m.cuda()


@pytest.mark.parametrize("cfg_name", sorted(CONFIGS))
def test_grouping_structure_full_size(cfg_name):
    """Every window: the first cnt slots ascend strictly (points are taken in index order), lie inside the window
    (|z - zc| < dis_z, strict, fp32), the remaining slots repeat the first hit, cnt = min(hits, nsample) against a
    brute-force fp32 count, empty windows are all-zero."""
    from frustum_convnet_amd.query_depth_point import query_depth_point
    data = _full_batch(cfg_name=cfg_name)
    pc = data["point_cloud"][:, :3].contiguous()
    z = pc[:, 2, :]
    for s, (dist, K) in enumerate(zip(CONFIGS[cfg_name][0], CONFIGS[cfg_name][3])):
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


@pytest.mark.parametrize("cfg_name", sorted(CONFIGS))
def test_full_step_is_bitwise_reproducible(cfg_name):
    """Two runs of the same full-size step give identical logits and identical gradients, bit for bit (no atomics on
    activations or gradients; BN sums are fp64 atomics of per-tile fp32 partials)."""
    data = _full_batch(cfg_name=cfg_name)
    outs = []
    for _ in range(2):
        m = _model(cfg_name=cfg_name)
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


@pytest.mark.parametrize("cfg_name", ["car", "sunrgbd"])
def test_inference_with_pool_keys_equals_pooling_pass_full_size(cfg_name):
    """The inference path takes the max-pool in conv3's epilogue (keys, y3 never written); FCN_POOL_KEYS=0 keeps the pooling pass
    over the stored y3.  Same model, same full-size batch: the logits must agree BIT FOR BIT (the pooled features are the same
    fp32 expression of the same winning row), and so must the training-mode forward with the keys forced on."""
    import os
    data = _full_batch(cfg_name=cfg_name)
    keys = [k for k in data if k in ("point_cloud", "one_hot") or k.startswith("center_ref")]
    out = {}
    try:
        for mode in ("default", "0", "1"):
            if mode == "default":
                os.environ.pop("FCN_POOL_KEYS", None)
            else:
                os.environ["FCN_POOL_KEYS"] = mode
            m = _model(cfg_name=cfg_name)
            m.eval()
            with torch.no_grad():
                m({k: data[k] for k in keys})
                ev = [t.clone() for t in m.last_logits]
            uses_keys = [w.pkey is not None for n in m.feat_net.nets for lst in n._pool.free.values() for w in lst]
            assert uses_keys and all(u == (mode != "0") for u in uses_keys), (mode, uses_keys)
            m.train()
            with torch.no_grad():
                m(data)
                tr = [t.clone() for t in m.last_logits]
            out[mode] = ev + tr
    finally:
        os.environ.pop("FCN_POOL_KEYS", None)
    for mode in ("default", "1"):
        for a, b in zip(out[mode], out["0"]):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("cfg_name", ["car", "sunrgbd"])
def test_train_statistics_are_the_only_coupling_between_frustums(cfg_name):
    """Training-mode BatchNorm couples the frustums ONLY through the batch statistics: duplicating the whole batch
    (2B frustums) leaves mean / biased variance unchanged, hence the logits of the first copy (1e-4)."""
    data = _full_batch(cfg_name=cfg_name)
    half = {k: v[:8].contiguous() for k, v in data.items()}
    dup = {k: torch.cat([v, v], 0) for k, v in half.items()}
    m1, m2 = _model(cfg_name=cfg_name), _model(cfg_name=cfg_name)
    m1.train()
    m2.train()
    with torch.no_grad():
        m1(half)
        c1, r1 = [t.clone() for t in m1.last_logits]
        m2(dup)
        c2, r2 = m2.last_logits
    assert float((c2[:8] - c1).abs().max()) < 1e-4 and float((c2[8:] - c1).abs().max()) < 1e-4
    assert float((r2[:8] - r1).abs().max()) < 1e-4 and float((r2[8:] - r1).abs().max()) < 1e-4
