"""ROUND-2 SCOUTING (not on the product path): conv3 of each PointNet scale with bf16 MFMA operands (fp32 accumulate) against
the product's exact-fp32 MFMA kernel on the bench batch: time per launch and error of the pre-BN output."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch
import bench
from frustum_convnet_amd import synth, _native, pointnet_fused as pf

so = os.path.join(HERE, "libscout.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(HERE, "bf16_conv3.hip"), "-o", so])
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
L = _native.lib()
S = ctypes.CDLL(so)
S.scout_conv3_bf16.restype = ctypes.c_int
S.scout_conv3_bf16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
xyz = data["point_cloud"][:, :3].contiguous()
nets = (model.feat_net.pointnet1, model.feat_net.pointnet2, model.feat_net.pointnet3, model.feat_net.pointnet4)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for s, net in enumerate(nets):
    ref = data["center_ref%d" % (s + 1)].contiguous()
    params, bufs = net._param_pack()
    cfgt = (float(net.dist), int(net.nsample), True, 1e-5, 0.1)
    with torch.no_grad():
        feat, idx, cnt, ws, desc, keep = pf._forward_impl(net._pool, cfgt, xyz, ref, None, bufs, params, False)
    pstruct = pf._params_struct(keep[0], keep[1], keep[2], [None] * 3, [None] * 3, [None] * 3)
    B, Lw, K, C1, C2, C3 = desc.B, desc.L, desc.K, desc.C1, desc.C2, desc.C3
    if C3 % 128:
        continue
    st = _native.current_stream(dev)
    y3_ref = ws.y3.clone()
    out = torch.zeros_like(ws.y3)
    bn2 = ws.bn[4 * C1:4 * C1 + 4 * C2]
    W3 = keep[0][2]
    f32 = lambda: _native.check(L.fcn_pn_conv_fwd(ctypes.byref(desc), ctypes.byref(pstruct), ctypes.byref(ws.c), 3, 0, st), "conv")
    b16 = lambda: S.scout_conv3_bf16(ws.woff.data_ptr(), ws.tiles.data_ptr(), ws.y2.data_ptr(), bn2.data_ptr(), W3.data_ptr(),
                                     out.data_ptr(), B, Lw, K, C2, C3, st)
    assert b16() == 0
    torch.cuda.synchronize()
    E = int(ws.woff[:, -1].sum())
    nent = ws.woff[:, -1].long()
    live = (torch.arange(Lw * K, device=dev)[None, :] < nent[:, None])
    d = (out - y3_ref)[live]
    r = y3_ref[live]
    t32, t16 = timeit(f32), timeit(b16)
    fl = 2.0 * E * C2 * C3
    print("scale %d conv3 %dx%d rows %d: fp32 MFMA %.1f us (%.0f TF)  bf16 MFMA %.1f us (%.0f TF)  speedup %.2fx | "
          "y3 max|err| %.3e rms err %.3e (rms y3 %.3e, max|y3| %.3e)"
          % (s + 1, C2, C3, E, t32, fl / t32 / 1e6, t16, fl / t16 / 1e6, t32 / t16, float(d.abs().max()),
             float(d.pow(2).mean().sqrt()), float(r.pow(2).mean().sqrt()), float(r.abs().max())))
    net._pool.release(ws)
