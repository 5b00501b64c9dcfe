"""Stage-by-stage comparison of the HIP PointNet path with the CPU emulation (tests/entry_ref.py) and the
dense oracle (oracle/det_ref.py).  Used by the -m gpu tests and runnable standalone for a diff table:

    python tests/gpu_stage_check.py            # prints max-abs-diff per stage for a few shapes
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import det_ref, grouping  # noqa: E402
import entry_ref  # noqa: E402
from frustum_convnet_amd import synth  # noqa: E402


def make_case(B, N, stride, K, mlp, dist, seed=5, variant="car", span=70.0):
    d = synth.make_batch(B, N, strides=(stride,) * 4, max_depth=span, seed=seed, variant=variant,
                         tilt=(0.01, 0.05))
    pc = torch.from_numpy(d["point_cloud"])
    ref = torch.from_numpy(d["center_ref1"])
    sd = {}
    cin = 3
    for j, co in enumerate(mlp):
        p = "m.conv%d" % (j + 1)
        sd[p + ".0.weight"] = torch.zeros(co, cin, 1, 1)
        sd[p + ".1.weight"] = torch.zeros(co)
        sd[p + ".1.bias"] = torch.zeros(co)
        sd[p + ".1.running_mean"] = torch.zeros(co)
        sd[p + ".1.running_var"] = torch.ones(co)
        sd[p + ".1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        cin = co
    synth.fill_state_dict(sd, seed=seed)
    one_hot = torch.zeros(B, 3)
    one_hot[:, 1] = 1.0
    return pc, ref, sd, one_hot


def _emu_forward(pf, cfgt, pcg, refg, one_hot, bufs, plist):
    """The launch sequence of query_depth_point + pointnet_fused._forward_impl on CPU tensors: used when the library behind
    _native.lib() is the host emulation of the kernels (tests/host_harness/build_emu.py), which takes host pointers."""
    import ctypes
    from frustum_convnet_amd import _native
    L = _native.lib()
    dist, K = cfgt[0], cfgt[1]
    b, _, n = pcg.shape
    m = refg.size(2)
    idx = torch.empty((b, m, K), dtype=torch.int64)
    cnt = torch.empty((b, m), dtype=torch.int32)
    _native.check(L.fcn_query_depth_point_f32(pcg.data_ptr() + 4 * 2 * n, 1, 3 * n, refg.data_ptr() + 4 * 2 * m, 1, 3 * m,
                                              b, n, m, float(dist), int(K), idx.data_ptr(), cnt.data_ptr(), None),
                  "fcn_query_depth_point_f32")
    h = pf._acquire(pf.WorkspacePool(), cfgt, pcg, refg, one_hot, bufs, plist, True)
    _native.check(L.fcn_pn_compact(ctypes.byref(h["desc"]), pcg.data_ptr(), refg.data_ptr(), idx.data_ptr(), cnt.data_ptr(),
                                   ctypes.byref(h["ws"].c), None), "fcn_pn_compact")
    Bq, Lw, C3 = h["dims"]
    feat = torch.empty((Bq, C3 + h["nvec"], Lw), dtype=torch.float32)
    _native.check(L.fcn_pn_forward(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), cnt.data_ptr(),
                                   None if h["oh"] is None else h["oh"].data_ptr(), ctypes.byref(h["ws"].c),
                                   feat.data_ptr(), None), "fcn_pn_forward")
    return idx, cnt, (feat, idx, cnt, h["ws"], h["desc"], (h["Wc"], h["gs"], h["bs"], cnt, idx, h["oh"]))


def run_stages(B, N, stride, K, mlp, dist, seed=5, variant="car", verbose=True, emu=False):
    """Returns dict stage -> (max abs diff, scale) comparing HIP vs CPU emulation / oracle.  emu: the library behind
    _native.lib() is the host emulation of the kernels -- same checks on CPU tensors, no stream, no synchronisation."""
    from frustum_convnet_amd import pointnet_fused as pf
    from frustum_convnet_amd.query_depth_point import query_depth_point

    dev = torch.device("cpu") if emu else torch.device("cuda:0")
    pc, ref, sd, one_hot = make_case(B, N, stride, K, mlp, dist, seed, variant)
    L = ref.shape[2]
    res = {}

    def rec(name, got, exp, exact=False):
        got = got.detach().cpu()
        exp = exp.detach().cpu()
        if got.shape != exp.shape:
            res[name] = (float("inf"), 0.0)
            if verbose:
                print("%-14s SHAPE MISMATCH %s vs %s" % (name, tuple(got.shape), tuple(exp.shape)))
            return
        if exact:
            d = float((got != exp).sum())
            sc = float(exp.numel())
        else:
            d = float((got.double() - exp.double()).abs().max()) if got.numel() else 0.0
            sc = float(exp.double().abs().max()) if exp.numel() else 0.0
        res[name] = (d, sc)
        if verbose:
            print("%-14s diff %.3e   (scale %.3e)%s" % (name, d, sc, "  [count of mismatches]" if exact else ""))

    # ---- oracle side (CPU)
    idx_np, cnt_np = grouping.query_depth_point(dist, K, pc.numpy(), ref.numpy())
    idx_o, cnt_o = torch.from_numpy(idx_np), torch.from_numpy(cnt_np)
    c = entry_ref.compact(idx_o, cnt_o, pc, ref, K)
    W = [sd["m.conv%d.0.weight" % j].view(mlp[j - 1], -1) for j in (1, 2, 3)]
    G = [sd["m.conv%d.1.weight" % j] for j in (1, 2, 3)]
    Bt = [sd["m.conv%d.1.bias" % j] for j in (1, 2, 3)]
    f = entry_ref.forward(c, W[0], G[0], Bt[0], W[1], G[1], Bt[1], W[2], G[2], Bt[2])
    dfeat = torch.from_numpy(synth.normalish(3, 1, (B, mlp[2] + 3, L)).astype(np.float32))
    r = entry_ref.backward(c, f, dfeat[:, :mlp[2]], W[0], G[0], W[1], G[1], W[2], G[2])

    # ---- HIP side
    pcg, refg = pc.to(dev), ref.to(dev)

    sdg = {k: v.clone().to(dev) for k, v in sd.items()}
    plist = []
    for j in (1, 2, 3):
        plist += [sdg["m.conv%d.0.weight" % j].requires_grad_(True), sdg["m.conv%d.1.weight" % j].requires_grad_(True),
                  sdg["m.conv%d.1.bias" % j].requires_grad_(True)]
    bufs = ([sdg["m.conv%d.1.running_mean" % j] for j in (1, 2, 3)],
            [sdg["m.conv%d.1.running_var" % j] for j in (1, 2, 3)],
            [sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)])
    pool = pf.WorkspacePool()
    cfgt = (float(dist), int(K), True, 1e-5, 0.1)
    if emu:
        idx_g, cnt_g, (feat, idx2, cnt2, ws, desc, keep) = _emu_forward(pf, cfgt, pcg, refg, one_hot, bufs, plist)
    else:
        idx_g, cnt_g = query_depth_point(dist, K, pcg, refg)
        feat, idx2, cnt2, ws, desc, keep = pf._forward_impl(pool, cfgt, pcg, refg, one_hot.to(dev), bufs, plist, True)
        torch.cuda.synchronize()
    rec("idx", idx_g, idx_o, exact=True)
    rec("cnt", cnt_g, cnt_o, exact=True)
    nent = c["nent"]
    rec("woff", ws.woff, c["woff"], exact=True)

    def live(t):   # (B,cap,...) -> (E,...)
        t = t.detach().cpu()
        return torch.cat([t[b, :int(nent[b])] for b in range(B)], 0)

    rec("ent", live(ws.ent), torch.cat([f["u"], f["w"][:, None]], 1))
    rec("ewin", live(ws.ewin), live(c["ewin"]), exact=True)
    C1, C2, C3 = mlp
    bn = ws.bn.detach().cpu()
    o2, o3 = 4 * C1, 4 * (C1 + C2)
    rec("bn1.scale", bn[0:C1], f["s"][0])
    rec("bn1.shift", bn[C1:2 * C1], f["t"][0])
    rec("y2", live(ws.y2), f["y2"])
    rec("bn2.mean", bn[o2 + 2 * C2:o2 + 3 * C2], f["mean"][1].float())
    rec("bn2.rstd", bn[o2 + 3 * C2:o2 + 4 * C2], f["rstd"][1].float())
    rec("y3", live(ws.y3), f["y3"])
    rec("bn3.mean", bn[o3 + 2 * C3:o3 + 3 * C3], f["mean"][2].float())
    rec("bn3.rstd", bn[o3 + 3 * C3:o3 + 4 * C3], f["rstd"][2].float())
    rec("feat", feat[:, :C3], f["feat"])
    rec("onehot", feat[:, C3:], one_hot[:, :, None].expand(-1, -1, L))
    # amax may legitimately differ on exact ties; compare the value at the argmax instead
    am_g = ws.amax.detach().cpu()
    rec("amax(-1 set)", (am_g < 0), (f["amax"] < 0), exact=True)
    # running stats
    M = float(B * L * K)
    for j in (1, 2, 3):
        exp_rm = 0.9 * sd["m.conv%d.1.running_mean" % j] + 0.1 * f["mean"][j - 1].float()
        exp_rv = 0.9 * sd["m.conv%d.1.running_var" % j] + 0.1 * f["var"][j - 1].float() * (M / (M - 1))
        rec("rmean%d" % j, sdg["m.conv%d.1.running_mean" % j], exp_rm)
        rec("rvar%d" % j, sdg["m.conv%d.1.running_var" % j], exp_rv)
    rec("nbt", torch.stack([sdg["m.conv%d.1.num_batches_tracked" % j] for j in (1, 2, 3)]),
        torch.ones(3, dtype=torch.int64), exact=True)

    # ---- backward through the C-ABI
    import ctypes
    from frustum_convnet_amd import _native
    Wc, gs, bs = keep[0], keep[1], keep[2]
    dW = [torch.empty_like(w) for w in Wc]
    dg = [torch.empty_like(g) for g in gs]
    db = [torch.empty_like(b) for b in bs]
    params = pf._params_struct(Wc, gs, bs, [None] * 3, [None] * 3, [None] * 3)
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    dfg = dfeat.to(dev).contiguous()
    rc = _native.lib().fcn_pn_backward(ctypes.byref(desc), ctypes.byref(params), dfg.data_ptr(), ctypes.byref(ws.c),
                                       arr(dW), arr(dg), arr(db), None if emu else _native.current_stream(dev))
    assert rc == 0, rc
    if not emu:
        torch.cuda.synchronize()
    if ws.dy3 is not None:          # (FCN_STORE_DY3=0: conv3's weight-gradient GEMM rebuilds dy3 and nothing stores it)
        rec("dy3", live(ws.dy3), r["dy3"])
    zmask2 = (f["y2"] * f["s"][1] + f["t"][1] > 0).float()
    rec("dz2", live(ws.dz2), r["G2"] * zmask2)
    for j in (3, 2, 1):
        rec("dW%d" % j, dW[j - 1], r["dW%d" % j])
        rec("dgamma%d" % j, dg[j - 1], r["dg%d" % j])
        rec("dbeta%d" % j, db[j - 1], r["db%d" % j])
    if not emu:
        pool.release(ws)

    # ---- dense oracle cross-check of the pooled features (independent of entry_ref)
    g, _, _ = det_ref.pointnet_module(pc, ref, sd, "m", dist, K, True, None, group=(idx_np, cnt_np))
    rec("feat~dense", feat[:, :C3], g.max(-1)[0])
    return res


# (B, N, stride, K, mlp, dist): small + the four real scales of the car config at B=4, N=512
CASES = [
    (2, 128, 3.5, 16, (64, 64, 128), 1.0),
    (3, 200, 2.5, 32, (64, 64, 128), 0.7),
    (4, 512, 0.25, 32, (64, 64, 128), 0.25),
    (4, 512, 0.5, 64, (64, 64, 128), 0.5),
    (4, 512, 1.0, 64, (128, 128, 256), 1.0),
    (4, 512, 2.0, 128, (256, 256, 512), 2.0),
]


def check(res, tol=2e-4):
    bad = []
    for k, (d, sc) in res.items():
        if k in ("idx", "cnt", "woff", "ewin", "nbt", "amax(-1 set)"):
            ok = d == 0
        else:
            ok = d <= tol * max(sc, 1e-3) + 1e-6
        if not ok:
            bad.append((k, d, sc))
    return bad


if __name__ == "__main__":
    torch.manual_seed(0)
    cases = CASES
    for cs in cases:
        print("==== case B=%d N=%d stride=%s K=%d mlp=%s dist=%s" % cs)
        try:
            res = run_stages(*cs)
            bad = check(res)
            print("   -> %s" % ("OK" if not bad else "FAIL: %s" % bad))
        except Exception as e:  # noqa
            import traceback
            traceback.print_exc()
