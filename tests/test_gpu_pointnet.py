"""-m gpu: every stage of the HIP PointNet scale (through the C-ABI) against the CPU emulation of the
entry-space algorithm (tests/entry_ref.py, itself proven equal to the dense oracle on CPU) and against the
dense oracle's pooled features.  Tolerance: 2e-4 of each tensor's max magnitude (fp32 summation order)."""
import pytest
import torch

import gpu_stage_check as gsc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gsc.CASES, ids=lambda c: "B%d_N%d_s%s_K%d_C%d" % (c[0], c[1], c[2], c[3], c[4][2]))
def test_stages(case):
    res = gsc.run_stages(*case, verbose=True)
    bad = gsc.check(res)
    assert not bad, bad


def test_uniform_variant_full_windows():
    res = gsc.run_stages(2, 512, 2.0, 8, (64, 64, 128), 2.0, variant="uniform")
    assert not gsc.check(res)


# (the last case: ~100 one- and two-row windows per 128-row tile -- more than the epilogue's LDS table has window slots)
@pytest.mark.parametrize("case", gsc.CASES[1:] + [(2, 128, 0.25, 16, (64, 64, 128), 0.3)],
                         ids=lambda c: "B%d_N%d_s%s_K%d_C%d" % (c[0], c[1], c[2], c[3], c[4][2]))
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_key_pool_matches_row_pool(case, training):
    """The max-pool taken in conv3's epilogue (keys + pool_keys_kernel) against the pooling pass over the stored y3: features
    bit for bit; the arg-max rows too, except where two rows of a window reach the same relu(bn(y)) (a rounding tie -- either
    row is 'the first maximum' of a different but equally valid evaluation order; the value at both must agree exactly).
    One gamma of conv3's BatchNorm is negative and one is zero: the keys orient the maximum by the sign of gamma."""
    import os
    from frustum_convnet_amd import pointnet_fused as pf
    from frustum_convnet_amd.precision import CODES

    B, N, stride, K, mlp, dist = case
    dev = torch.device("cuda:0")
    pc, ref, sd, one_hot = gsc.make_case(B, N, stride, K, mlp, dist)
    sd["m.conv3.1.weight"][1] = -0.7
    sd["m.conv3.1.weight"][2] = 0.0
    out = {}
    for keys in ("1", "0"):
        os.environ["FCN_POOL_KEYS"] = keys
        sdg = {k: v.clone().to(dev) for k, v in sd.items()}
        plist = []
        for j in (1, 2, 3):
            plist += [sdg["m.conv%d.0.weight" % j], sdg["m.conv%d.1.weight" % j], sdg["m.conv%d.1.bias" % j]]
        bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)], [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
                [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
        pool = pf.WorkspacePool()
        cfgt = (float(dist), int(K), training, 1e-5, 0.1, True, True)          # (.., need_grad, nlc)
        for rep in range(2):                # twice on the same workspace: the keys are back at zero after a forward
            feat, idx, cnt, ws, desc, keep = pf._forward_impl(pool, cfgt, pc.to(dev), ref.to(dev), None, bufs, plist, True)
            if rep == 0:
                pool.release(ws)
        assert (ws.pkey is not None) == (keys == "1")
        if ws.pkey is not None:
            assert int((ws.pkey != 0).sum()) == 0
        y3 = pf.Workspace.stored(ws.y3, CODES["split"]).detach().cpu().clone()
        for b in range(B):                  # (rows past a frustum's live entries are never written)
            y3[b, int(ws.woff[b, -1]):] = 0
        out[keys] = (feat.detach().cpu(), ws.amax.detach().cpu(), ws.bn.detach().cpu(), y3, bufs[0][2].detach().cpu())
    del os.environ["FCN_POOL_KEYS"]
    fk, ak, bnk, y3k, rmk = out["1"]
    fr, ar, bnr, y3r, rmr = out["0"]
    assert torch.equal(bnk, bnr) and torch.equal(rmk, rmr)
    assert not training or torch.equal(y3k, y3r)          # (eval mode with keys: y3 is not written at all)
    assert torch.equal(fk, fr)
    if not training:                    # (no arg-max in eval mode)
        return
    assert torch.equal(ak < 0, ar < 0)
    diff = (ak != ar).nonzero()
    C3 = mlp[2]
    s, t = bnk[4 * (mlp[0] + mlp[1]):][:C3], bnk[4 * (mlp[0] + mlp[1]):][C3:2 * C3]
    for b, l, c in diff.tolist():
        zk = torch.relu(torch.addcmul(t[c], s[c], y3k[b, ak[b, l, c], c]))
        zr = torch.relu(torch.addcmul(t[c], s[c], y3k[b, ar[b, l, c], c]))
        assert float(zk) == float(zr), (b, l, c, float(zk), float(zr))
    assert len(diff) <= max(4, ak.numel() // 10000), len(diff)


@pytest.mark.parametrize("case", [(3, 200, 2.5, 32, (64, 64, 128), 0.7), (4, 512, 2.0, 128, (256, 256, 512), 2.0)],
                         ids=lambda c: "B%d_C%d-K%d" % (c[0], c[4][2], c[3]))
def test_key_pool_backward_with_zero_and_negative_gamma(case):
    """Backward behind a key-pooled training forward against the one behind the row pooling, with one channel of BN3 at gamma = 0 and
    beta > 0 (its pooled feature is relu(beta) > 0, every row of a window ties, the FIRST row wins and d gamma = sum dz * xhat of
    THAT row -- the key of such a channel carries no value, so pool_keys_kernel fetches the winner's y3 itself: ADVICE r5) and one at
    gamma < 0.  Every gradient of the scale agrees to 1e-5 of its maximum (arg-max rounding ties aside, the two are the same sums)."""
    import ctypes
    import os
    import numpy as np
    from frustum_convnet_amd import _native, pointnet_fused as pf, synth

    B, N, stride, K, mlp, dist = case
    dev = torch.device("cuda:0")
    pc, ref, sd, one_hot = gsc.make_case(B, N, stride, K, mlp, dist)
    sd["m.conv3.1.weight"][1] = -0.7
    sd["m.conv3.1.weight"][2] = 0.0
    sd["m.conv3.1.bias"][2] = 0.5
    L = ref.shape[2]
    dfeat = torch.from_numpy(synth.normalish(3, 1, (B, L, mlp[2])).astype(np.float32)).to(dev).contiguous()       # position-major
    out = {}
    old = os.environ.get("FCN_POOL_KEYS")
    try:
        for keys in ("1", "0"):
            os.environ["FCN_POOL_KEYS"] = keys
            sdg = {k: v.clone().to(dev) for k, v in sd.items()}
            plist = []
            for j in (1, 2, 3):
                plist += [sdg["m.conv%d.0.weight" % j], sdg["m.conv%d.1.weight" % j], sdg["m.conv%d.1.bias" % j]]
            bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)], [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
                    [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
            pool = pf.WorkspacePool()
            cfgt = (float(dist), int(K), True, 1e-5, 0.1, True, True)          # (.., need_grad, nlc)
            feat, idx, cnt, ws, desc, keep = pf._forward_impl(pool, cfgt, pc.to(dev), ref.to(dev), None, bufs, plist, True)
            assert (ws.pkey is not None) == (keys == "1")
            assert float(feat[:, :, 2].max()) > 0          # the gamma = 0 channel pools to relu(beta) on every live window
            Wc, gs, bs = keep[0], keep[1], keep[2]
            dW = [torch.empty_like(w) for w in Wc]
            dg = [torch.empty_like(g) for g in gs]
            db = [torch.empty_like(b) for b in bs]
            params = pf._params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
            arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
            rc = _native.lib().fcn_pn_backward(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(), ctypes.byref(ws.c),
                                               arr(dW), arr(dg), arr(db), _native.current_stream(dev))
            assert rc == 0, rc
            torch.cuda.synchronize()
            out[keys] = [t.detach().cpu() for t in dW + dg + db]
    finally:
        if old is None:
            os.environ.pop("FCN_POOL_KEYS", None)
        else:
            os.environ["FCN_POOL_KEYS"] = old
    names = ["dW1", "dW2", "dW3", "dg1", "dg2", "dg3", "db1", "db2", "db3"]
    for n, a, b in zip(names, out["1"], out["0"]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, (n, float((a - b).abs().max()), float(b.abs().max()))
    dg3k, dg3r = out["1"][5], out["0"][5]
    assert abs(float(dg3r[2])) > 0                       # the channel really has a d gamma
    assert abs(float(dg3k[2] - dg3r[2])) <= 1e-5 * abs(float(dg3r[2])) + 1e-7, (float(dg3k[2]), float(dg3r[2]))


@pytest.mark.parametrize("case", [gsc.CASES[1], gsc.CASES[4], gsc.CASES[5]], ids=lambda c: "B%d_N%d_s%s_K%d_C%d" % (c[0], c[1], c[2], c[3], c[4][2]))
def test_rebuilt_dy3_gives_bit_identical_gradients(case):
    """FCN_STORE_DY3=0: dy3 is never materialised -- conv3's weight-gradient GEMM rebuilds it while staging from what the
    data-gradient GEMM reads (y3, arg-max / routed-gradient maps, BN3-backward sums); the default keeps the buffer and the
    read-back.  Same fp32 expression on the same inputs: every gradient of the scale must agree BIT FOR BIT."""
    import ctypes
    import os
    import numpy as np
    from frustum_convnet_amd import _native, pointnet_fused as pf, synth

    B, N, stride, K, mlp, dist = case
    dev = torch.device("cuda:0")
    pc, ref, sd, one_hot = gsc.make_case(B, N, stride, K, mlp, dist)
    L = ref.shape[2]
    dfeat = torch.from_numpy(synth.normalish(3, 1, (B, mlp[2] + 3, L)).astype(np.float32)).to(dev).contiguous()
    out = {}
    try:
        for mode in ("0", "1"):
            os.environ["FCN_STORE_DY3"] = mode
            sdg = {k: v.clone().to(dev) for k, v in sd.items()}
            plist = []
            for j in (1, 2, 3):
                plist += [sdg["m.conv%d.0.weight" % j], sdg["m.conv%d.1.weight" % j], sdg["m.conv%d.1.bias" % j]]
            bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)], [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
                    [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
            pool = pf.WorkspacePool()
            feat, idx, cnt, ws, desc, keep = pf._forward_impl(pool, (float(dist), int(K), True, 1e-5, 0.1), pc.to(dev), ref.to(dev),
                                                              one_hot.to(dev), bufs, plist, True)
            assert (ws.dy3 is not None) == (mode == "1")
            Wc, gs, bs = keep[0], keep[1], keep[2]
            dW = [torch.empty_like(w) for w in Wc]
            dg = [torch.empty_like(g) for g in gs]
            db = [torch.empty_like(b) for b in bs]
            params = pf._params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
            arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
            rc = _native.lib().fcn_pn_backward(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(), ctypes.byref(ws.c),
                                               arr(dW), arr(dg), arr(db), _native.current_stream(dev))
            assert rc == 0, rc
            torch.cuda.synchronize()
            out[mode] = [t.detach().cpu() for t in dW + dg + db]
    finally:
        os.environ.pop("FCN_STORE_DY3", None)
    for a, b in zip(out["0"], out["1"]):
        assert torch.equal(a, b)
    assert float(out["0"][2].abs().max()) > 0


def _scale_backward(case, env, two_streams=False):
    """Forward + backward of one scale through the C-ABI under the environment `env` -> (gradients on the CPU, workspace)."""
    import ctypes
    import os
    import numpy as np
    from frustum_convnet_amd import _native, pointnet_fused as pf, synth

    B, N, stride, K, mlp, dist = case
    dev = torch.device("cuda:0")
    pc, ref, sd, one_hot = gsc.make_case(B, N, stride, K, mlp, dist)
    L = ref.shape[2]
    dfeat = torch.from_numpy(synth.normalish(3, 1, (B, mlp[2] + 3, L)).astype(np.float32)).to(dev).contiguous()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sdg = {k: v.clone().to(dev) for k, v in sd.items()}
        plist = []
        for j in (1, 2, 3):
            plist += [sdg["m.conv%d.0.weight" % j], sdg["m.conv%d.1.weight" % j], sdg["m.conv%d.1.bias" % j]]
        bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)], [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
                [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
        pool = pf.WorkspacePool()
        feat, idx, cnt, ws, desc, keep = pf._forward_impl(pool, (float(dist), int(K), True, 1e-5, 0.1), pc.to(dev), ref.to(dev),
                                                          one_hot.to(dev), bufs, plist, True)
        Wc, gs, bs = keep[0], keep[1], keep[2]
        dW = [torch.empty_like(w) for w in Wc]
        dg = [torch.empty_like(g) for g in gs]
        db = [torch.empty_like(b) for b in bs]
        params = pf._params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        if two_streams:
            side, _evs, evarr, _s3 = pool.side_stream(dev)
            rc = _native.lib().fcn_pn_backward2(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(), ctypes.byref(ws.c),
                                                arr(dW), arr(dg), arr(db), _native.current_stream(dev),
                                                ctypes.c_void_p(side.cuda_stream), evarr)
        else:
            rc = _native.lib().fcn_pn_backward(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(), ctypes.byref(ws.c),
                                               arr(dW), arr(dg), arr(db), _native.current_stream(dev))
        assert rc == 0, rc
        torch.cuda.synchronize()
        return [t.detach().cpu() for t in dW + dg + db], ws
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("case", [(3, 200, 2.5, 32, (64, 64, 128), 0.7), (4, 512, 1.0, 64, (128, 128, 256), 1.0),
                                  (4, 512, 2.0, 128, (256, 256, 512), 2.0), (2, 130, 0.5, 64, (128, 128, 256), 0.4)],
                         ids=lambda c: "B%d_C%d-K%d" % (c[0], c[4][2], c[3]))
@pytest.mark.parametrize("two_streams", [False, True], ids=["one_stream", "two_streams"])
def test_tail_launch_gives_bit_identical_gradients(case, two_streams):
    """fcn_pn_ws.partial_both = 2: both weight gradients' reduces (and, on one stream, the layer-1 finalisation) as roles of ONE launch
    at the tail of the chain (pn_tail_kernel) against GEMM and reduce in alternation (partial_both = 0).  The same reduce on the
    same partials: every gradient of the scale agrees BIT FOR BIT, on one stream and with the weight gradients on a second."""
    ref, ws0 = _scale_backward(case, {"FCN_PN_TAIL": "0", "FCN_PN_MID": "0"}, two_streams)
    got, ws1 = _scale_backward(case, {"FCN_PN_TAIL": "1", "FCN_PN_MID": "0"}, two_streams)
    assert int(ws0.c.partial_both) == 0 and int(ws1.c.partial_both) == 2
    for a, b in zip(ref, got):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(3, 200, 2.5, 32, (64, 64, 128), 0.7), (4, 512, 1.0, 64, (128, 128, 256), 1.0),
                                  (4, 512, 2.0, 128, (256, 256, 512), 2.0)], ids=lambda c: "C%d-K%d" % (c[4][2], c[3]))
def test_merged_mid_launch_gives_bit_identical_gradients(case):
    """One-stream backward of a scale: conv2's data gradient and both weight-gradient GEMMs as roles of ONE launch
    (pn_mid_kernel, fcn_pn_ws.partial_both = 1) against the three launches one after the other (partial_both = 0).  The same
    kernels bodies on the same inputs, fixed-order reduces: every gradient of the scale agrees BIT FOR BIT."""
    import ctypes
    import os
    import numpy as np
    from frustum_convnet_amd import _native, pointnet_fused as pf, synth

    B, N, stride, K, mlp, dist = case
    dev = torch.device("cuda:0")
    pc, ref, sd, one_hot = gsc.make_case(B, N, stride, K, mlp, dist)
    L = ref.shape[2]
    dfeat = torch.from_numpy(synth.normalish(3, 1, (B, mlp[2] + 3, L)).astype(np.float32)).to(dev).contiguous()
    out = {}
    try:
        for mode in ("0", "1"):
            os.environ["FCN_PN_MID"] = mode
            os.environ["FCN_PN_TAIL"] = "0"
            sdg = {k: v.clone().to(dev) for k, v in sd.items()}
            plist = []
            for j in (1, 2, 3):
                plist += [sdg["m.conv%d.0.weight" % j], sdg["m.conv%d.1.weight" % j], sdg["m.conv%d.1.bias" % j]]
            bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)], [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
                    [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
            pool = pf.WorkspacePool()
            feat, idx, cnt, ws, desc, keep = pf._forward_impl(pool, (float(dist), int(K), True, 1e-5, 0.1), pc.to(dev), ref.to(dev),
                                                              one_hot.to(dev), bufs, plist, True)
            assert int(ws.c.partial_both) == int(mode)
            Wc, gs, bs = keep[0], keep[1], keep[2]
            dW = [torch.empty_like(w) for w in Wc]
            dg = [torch.empty_like(g) for g in gs]
            db = [torch.empty_like(b) for b in bs]
            params = pf._params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
            arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
            rc = _native.lib().fcn_pn_backward(ctypes.byref(desc), ctypes.byref(params), dfeat.data_ptr(), ctypes.byref(ws.c),
                                               arr(dW), arr(dg), arr(db), _native.current_stream(dev))
            assert rc == 0, rc
            torch.cuda.synchronize()
            out[mode] = [t.detach().cpu() for t in dW + dg + db]
    finally:
        os.environ.pop("FCN_PN_MID", None)
        os.environ.pop("FCN_PN_TAIL", None)
    for a, b in zip(out["0"], out["1"]):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(a, b)
