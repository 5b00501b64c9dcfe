"""-m gpu: the fused front (fcn_pn_group_compact: grouping + compaction + tile list + input moments + BN1 of all scales in
one launch, no int64 idx) against (1) the entry-space emulation applied to the ORACLE's idx (tests/entry_ref.compact over
oracle/grouping.py) -- exact -- and (2) the unfused C-ABI path fcn_query_depth_point_f32 + fcn_pn_compact + BN1 finalise."""
import ctypes

import numpy as np
import pytest
import torch

import entry_ref
from oracle import grouping
from frustum_convnet_amd import synth

pytestmark = pytest.mark.gpu
NS = (32, 64, 64, 128)
MLP = ((64, 64, 128), (64, 64, 128), (128, 128, 256), (256, 256, 512))


def _params(C, seed):
    g = torch.Generator().manual_seed(seed)
    W = [torch.randn(C[0], 3, generator=g) * 0.5, torch.randn(C[1], C[0], generator=g) * 0.1, torch.randn(C[2], C[1], generator=g) * 0.1]
    gam = [torch.rand(c, generator=g) + 0.5 for c in C]
    bet = [torch.randn(c, generator=g) * 0.1 for c in C]
    plist = []
    for i in range(3):
        plist += [W[i].cuda(), gam[i].cuda(), bet[i].cuda()]
    bufs = ([torch.zeros(c).cuda() for c in C], [torch.ones(c).cuda() for c in C],
            [torch.zeros((), dtype=torch.int64).cuda() for c in C])
    return plist, bufs


@pytest.mark.parametrize("B,N,strides,variant", [(4, 512, (0.25, 0.5, 1.0, 2.0), "car"), (3, 700, (0.1, 0.2, 0.4, 0.8), "uniform"),
                                                (32, 1024, (0.25, 0.5, 1.0, 2.0), "car"), (2, 130, (2.0, 2.0, 4.0, 8.0), "car")])
def test_group_compact_matches_oracle_and_unfused(B, N, strides, variant):
    from frustum_convnet_amd import pointnet_fused as pf, _native
    data = synth.make_batch(B, N, strides=strides, seed=77, variant=variant, tilt=(0.01, 0.05))
    pc = torch.from_numpy(data["point_cloud"]).cuda()
    pools = [pf.WorkspacePool() for _ in range(4)]
    handles, unf = [], []
    for s in range(4):
        ref = torch.from_numpy(data["center_ref%d" % (s + 1)]).cuda()
        plist, bufs = _params(MLP[s], 10 + s)
        cfgt = (float(strides[s]), NS[s], True, 1e-5, 0.1, False, True)
        handles.append(pf._acquire(pools[s], cfgt, pc, ref, None, bufs, plist, False))
        plist2, bufs2 = _params(MLP[s], 10 + s)
        unf.append((pf._acquire(pools[s], cfgt, pc, ref, None, bufs2, plist2, False), ref, bufs2, bufs))
    for rep in range(2):                # twice: the arrival counters must be left at zero
        pf.group_compact(handles, pc)
    torch.cuda.synchronize()
    L = _native.lib()
    for s in range(4):
        h = handles[s]
        K, Lw = NS[s], h["desc"].L
        ref_np = data["center_ref%d" % (s + 1)]
        # (1) oracle idx -> entry-space emulation: exact
        idx_o, cnt_o = grouping.query_depth_point(float(strides[s]), K, data["point_cloud"], ref_np)
        c = entry_ref.compact(torch.from_numpy(idx_o), torch.from_numpy(cnt_o), torch.from_numpy(data["point_cloud"]),
                              torch.from_numpy(ref_np), K)
        ws = h["ws"]
        assert torch.equal(ws.cnt.cpu(), torch.from_numpy(cnt_o)), s
        assert torch.equal(ws.woff.cpu(), c["woff"]), s
        ent, ewin = ws.ent.cpu(), ws.ewin.cpu()
        for b in range(B):
            n = int(c["nent"][b])
            assert torch.equal(ent[b, :n], c["ent"][b, :n]), (s, b)
            assert torch.equal(ewin[b, :n], c["ewin"][b, :n]), (s, b)
        # (2) unfused C-ABI path on a second workspace
        hu, ref, bufs_u, bufs_f = unf[s]
        idx, cnt = pf.query_depth_point(float(strides[s]), K, pc, ref)
        _native.check(L.fcn_pn_compact(ctypes.byref(hu["desc"]), pc.data_ptr(), ref.data_ptr(), idx.data_ptr(), cnt.data_ptr(),
                                       ctypes.byref(hu["ws"].c), _native.current_stream(pc.device)), "fcn_pn_compact")
        feat_u = pf._run_forward(hu, cnt, idx)[0]
        torch.cuda.synchronize()
        wu = hu["ws"]
        nt = int(wu.tiles[0])
        assert int(ws.tiles[0]) == nt and torch.equal(ws.tiles[4:4 + nt], wu.tiles[4:4 + nt]), s
        assert int(ws.tiles[1]) == 0 and float(ws.gmom.view(B, 12)[:, 10].abs().max()) == 0.0
        mom_f, mom_u = ws.stat[:10].cpu().numpy(), wu.stat[:10].cpu().numpy()
        assert np.allclose(mom_f, mom_u, rtol=1e-12, atol=1e-9), (s, mom_f, mom_u)
        C1 = MLP[s][0]
        assert torch.allclose(ws.bn[:4 * C1], wu.bn[:4 * C1], rtol=1e-5, atol=1e-6), s
        # running statistics of conv1's BN: updated twice by the fused path (two launches), once by the unfused one
        # (from zero: 0.1 * mean once, 0.9 * 0.1 * mean + 0.1 * mean = 0.19 * mean twice)
        assert torch.allclose(bufs_f[0][0], 1.9 * bufs_u[0][0], rtol=1e-4, atol=1e-6), s
        assert int(bufs_f[2][0]) == 2 and int(bufs_u[2][0]) == 1
        # and the forward on the grouped workspace gives the same pooled features
        feat_f = pf._run_forward(h, ws.cnt, pf._empty_idx(pc.device))[0]
        torch.cuda.synchronize()
        assert torch.allclose(feat_f, feat_u, rtol=1e-5, atol=1e-6), (s, float((feat_f - feat_u).abs().max()))


def test_fused_front_model_matches_unfused():
    """Whole model: fused_front on/off give the same logits (both against the golden bar) and the same gradients."""
    from test_gpu_model import _model
    from helpers import load_golden, golden_inputs
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    outs = []
    for ff in (True, False):
        m = _model(g)
        m.feat_net.fused_front = ff
        m.train()
        losses, _ = m(data)
        losses["total_loss"].backward()
        outs.append((torch.cat([t.flatten() for t in m.last_logits]).detach(), {n: p.grad.clone() for n, p in m.named_parameters()},
                     {k: v.clone() for k, v in m.state_dict().items()}))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-5
    for n in outs[0][1]:
        a, b = outs[0][1][n], outs[1][1][n]
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-7, n
    for k in outs[0][2]:
        if "running" in k or k.endswith("num_batches_tracked"):
            assert torch.allclose(outs[0][2][k].float(), outs[1][2][k].float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("B,N,strides", [(4, 512, (0.25, 0.5, 1.0, 2.0)), (32, 1024, (0.25, 0.5, 1.0, 2.0))])
def test_phased_front_is_bit_identical_to_the_fused_front(B, N, strides):
    """fcn_pn_group_compact2: phase 1 (batch-only part) + phase 2 (weight images + BN1 fold) leave every workspace buffer, the
    BN1 block and the running statistics bit-identical to the fused phase 3 -- also when the weights CHANGE between the two
    phases (the prefetch case: phase 1 runs before the optimiser step, phase 2 after it)."""
    from frustum_convnet_amd import pointnet_fused as pf
    data = synth.make_batch(B, N, strides=strides, seed=91, variant="car", tilt=(0.01, 0.05))
    pc = torch.from_numpy(data["point_cloud"]).cuda()
    pools = [pf.WorkspacePool() for _ in range(4)]
    fused, phased, params = [], [], []
    for s in range(4):
        ref = torch.from_numpy(data["center_ref%d" % (s + 1)]).cuda()
        cfgt = (float(strides[s]), NS[s], True, 1e-5, 0.1, False, True)
        pa, ba = _params(MLP[s], 20 + s)
        pb, bb = _params(MLP[s], 20 + s)
        fused.append(pf._acquire(pools[s], cfgt, pc, ref, None, ba, pa, False))
        phased.append(pf._acquire(pools[s], cfgt, pc, ref, None, bb, pb, False))
        params.append((pa, ba, pb, bb))
    pf.group_compact(phased, pc, phase=1)
    torch.cuda.synchronize()
    assert all(h["desc"].grouped == 0 for h in phased)
    for pa, ba, pb, bb in params:           # "the optimiser step": the same in-place update of both parameter sets
        for ta, tb in zip(pa, pb):
            ta.mul_(1.25).add_(0.01)
            tb.mul_(1.25).add_(0.01)
    pf.group_compact(phased, pc, phase=2)
    pf.group_compact(fused, pc)
    torch.cuda.synchronize()
    assert all(h["desc"].grouped == 1 for h in phased)
    for s in range(4):
        wf, wp = fused[s]["ws"], phased[s]["ws"]
        for name in ("cnt", "woff", "tiles", "stat", "bn", "wenc", "gmom"):
            a, b = getattr(wf, name), getattr(wp, name)
            if name == "bn":
                a, b = a[:4 * MLP[s][0]], b[:4 * MLP[s][0]]
            assert torch.equal(a, b), (s, name)
        for b_ in range(B):
            n = int(wf.woff[b_, -1])
            assert torch.equal(wf.ent[b_, :n], wp.ent[b_, :n]) and torch.equal(wf.ewin[b_, :n], wp.ewin[b_, :n]), (s, b_)
        pa, ba, pb, bb = params[s]
        assert torch.equal(ba[0][0], bb[0][0]) and torch.equal(ba[1][0], bb[1][0]) and int(ba[2][0]) == int(bb[2][0]) == 1, s
        ff = pf._run_forward(fused[s], wf.cnt, pf._empty_idx(pc.device))[0]
        fp = pf._run_forward(phased[s], wp.cnt, pf._empty_idx(pc.device))[0]
        torch.cuda.synchronize()
        assert torch.equal(ff, fp), s


def test_prefetched_front_gives_the_same_training_steps():
    """PointNetDet.prefetch(): three optimiser steps with the next batch's front prefetched beside the backward are bit-identical
    (logits, losses, parameters, running statistics) to three steps without it; a prefetch for a batch the next forward does not
    get is dropped."""
    from test_gpu_model import _model
    from helpers import load_golden, golden_inputs
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    other = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
    runs = []
    for pre in (False, True):
        m = _model(g)
        m.train()
        m.defer_metrics_join = pre
        st = FlatTrainState(m, lr=1e-3, weight_decay=1e-4)
        logs = []
        for step in range(3):
            losses, _ = m(data)
            if pre:
                assert m.prefetch(other if step == 1 else data)       # step 1 prefetches a batch nobody will pass: dropped
            m.backward(losses["total_loss"])
            st.adam_step()
            logs.append((torch.cat([t.flatten() for t in m.last_logits]).detach().clone(), losses["total_loss"].detach().clone()))
        m.feat_net.drop_prefetch()
        torch.cuda.synchronize()
        runs.append((logs, {k: v.clone() for k, v in m.state_dict().items()}))
    for (la, ta), (lb, tb) in zip(runs[0][0], runs[1][0]):
        assert torch.equal(la, lb) and torch.equal(ta, tb)
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_forward_weight_images_hold_the_fp16_split_bit_for_bit():
    """The fp16 x 3 operand encode (gemm_tile.h enc2: v_cvt_pk_f16_f32 + v_fma_mixlo / v_fma_mixhi) is hi = rne16(x),
    lo = rne16(x - hi) -- the arithmetic DESIGN.md states and the host emulation restates in C++: the conv2 / conv3 forward images
    packed on the GPU (pn_pack.h layout: [Cin/32][plane][4 k-blocks][Cout] u32x4, dword q = the pair (8kb + 2q, 8kb + 2q + 1))
    against numpy, bit for bit, on weights that cover fp16's normal, subnormal and overflow-free range, both signs, exact fp16
    values (lo = 0) and values whose residual is itself subnormal."""
    from frustum_convnet_amd import pointnet_fused as pf
    B, N, strides = 4, 512, (0.25, 0.5, 1.0, 2.0)
    data = synth.make_batch(B, N, strides=strides, seed=5, variant="car", tilt=(0.01, 0.05))
    pc = torch.from_numpy(data["point_cloud"]).cuda()
    pools = [pf.WorkspacePool() for _ in range(4)]
    handles, weights = [], []
    rng = np.random.default_rng(11)
    for s in range(4):
        ref = torch.from_numpy(data["center_ref%d" % (s + 1)]).cuda()
        plist, bufs = _params(MLP[s], 40 + s)
        for li in (1, 2):                       # conv2, conv3: magnitudes 2^-30 .. 2^15, a tenth of them exact fp16 numbers
            W = plist[3 * li]
            mag = np.exp2(rng.uniform(-30.0, 15.0, size=tuple(W.shape))).astype(np.float32)
            x = (mag * rng.choice([-1.0, 1.0], size=mag.shape)).astype(np.float32)
            exact = rng.random(mag.shape) < 0.1
            x[exact] = x[exact].astype(np.float16).astype(np.float32)
            x.flat[:4] = [0.0, -0.0, 65504.0, -6.0e-8]
            W.copy_(torch.from_numpy(x))
        weights.append([plist[3].cpu().numpy(), plist[6].cpu().numpy()])
        cfgt = (float(strides[s]), NS[s], True, 1e-5, 0.1, False, True)
        handles.append(pf._acquire(pools[s], cfgt, pc, ref, None, bufs, plist, False))
    pf.group_compact(handles, pc)
    torch.cuda.synchronize()
    for s in range(4):
        C1, C2, C3 = MLP[s]
        raw = handles[s]["ws"].wenc.cpu().numpy().view(np.uint16)           # 2 halves per float
        off = 0
        for W, (COUT, CIN) in zip(weights[s], ((C2, C1), (C3, C2))):
            img = raw[2 * off: 2 * (off + COUT * CIN)].reshape(CIN // 32, 2, 4, COUT, 8)      # [c][plane][kb][n][8 halves]
            off += COUT * CIN
            x = W.reshape(COUT, CIN // 32, 4, 8).transpose(1, 2, 0, 3)                         # [c][kb][n][j]
            hi = x.astype(np.float16)
            lo = (x - hi.astype(np.float32)).astype(np.float16)
            assert np.array_equal(img[:, 0], hi.view(np.uint16)), (s, COUT, "hi")
            assert np.array_equal(img[:, 1], lo.view(np.uint16)), (s, COUT, "lo")
