"""Isolated timings of the PointNet scale kernels at the bench shape (B=32, N=1024 car batch), HIP events, eager launches:
conv2 / conv3 GEMM alone (fcn_pn_conv_fwd), the whole forward of a scale (fcn_pn_forward) and forward + backward through
autograd.  FCN_LIB_NAME selects a tuning build.  Usage: python tools/pn_micro.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from frustum_convnet_amd import synth, _native, pointnet_fused as pf

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
model = bench.build_model(dev)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
xyz = data["point_cloud"][:, :3].contiguous()
refs = [data["center_ref%d" % i] for i in (1, 2, 3, 4)]
nets = model.feat_net.nets
L = _native.lib()


def timed(fn, n=reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
hs = [net.prepare_pooled(xyz, refs[s], data["one_hot"], True) for s, net in enumerate(nets)]
pf.group_compact(hs, xyz)
for s, (net, h) in enumerate(zip(nets, hs)):
    feat = torch.empty(h["dims"], dtype=torch.float32, device=dev)
    st = _native.current_stream(dev)
    full = lambda: _native.check(L.fcn_pn_forward(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), h["ws"].cnt.data_ptr(), None,
                                                  ctypes.byref(h["ws"].c), feat.data_ptr(), st), "fwd")
    full()
    c2 = lambda: _native.check(L.fcn_pn_conv_fwd(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), ctypes.byref(h["ws"].c), 2, 1, st), "c2")
    c3 = lambda: _native.check(L.fcn_pn_conv_fwd(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), ctypes.byref(h["ws"].c), 3, 1, st), "c3")
    t2, t3, tf = timed(c2), timed(c3), timed(full)
    out.append("scale %d: conv2 %6.1f  conv3 %6.1f  forward %6.1f us" % (s + 1, t2, t3, tf))
# whole backward chain of a scale on its saved forward state, straight through the C-ABI (single stream, eager)
from frustum_convnet_amd._native import c_fp
hs2 = [net.prepare_pooled(xyz, refs[s], data["one_hot"], True) for s, net in enumerate(nets)]    # (training, grad enabled)
assert all(h["ws"].dz2 is not None for h in hs2)
pf.group_compact(hs2, xyz)
for s, h in enumerate(hs2):
    feat = torch.empty(h["dims"], dtype=torch.float32, device=dev)
    st = _native.current_stream(dev)
    _native.check(L.fcn_pn_forward(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), h["ws"].cnt.data_ptr(), None,
                                   ctypes.byref(h["ws"].c), feat.data_ptr(), st), "fwd")
    dfeat = torch.randn_like(feat)
    C1, C2, C3 = h["desc"].C1, h["desc"].C2, h["desc"].C3
    dW = [torch.empty(C1 * 3, device=dev), torch.empty(C2 * C1, device=dev), torch.empty(C3 * C2, device=dev)]
    dg = [torch.empty(c, device=dev) for c in (C1, C2, C3)]
    db = [torch.empty(c, device=dev) for c in (C1, C2, C3)]
    arr = lambda ts: (c_fp * 3)(*[t.data_ptr() for t in ts])
    bw = lambda: _native.check(L.fcn_pn_backward2(ctypes.byref(h["desc"]), ctypes.byref(h["params"]), dfeat.data_ptr(),
                                                  ctypes.byref(h["ws"].c), arr(dW), arr(dg), arr(db), st, None, None), "bwd")
    out[s] += "  backward %6.1f us" % timed(bw)
print("lib %s" % os.environ.get("FCN_LIB_NAME", "libfcn_hip.so"))
print("\n".join(out))
sys.exit(0)
for s, net in enumerate(nets):
    dfeat = None

    def fb():
        global dfeat
        f = net.forward_pooled(xyz, refs[s], data["one_hot"], nlc=True)
        if dfeat is None:
            dfeat = torch.randn_like(f)
        f.backward(dfeat)
    tfb = timed(fb, max(reps // 2, 5))
    out[s] += "  fwd+bwd (unfused front, autograd) %6.1f us" % tfb
print("lib %s" % os.environ.get("FCN_LIB_NAME", "libfcn_hip.so"))
print("\n".join(out))
