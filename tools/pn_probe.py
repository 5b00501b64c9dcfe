"""Cycle accounting of the PointNet data-gradient GEMM (tuning build libfcn_hip_pnprobe.so, -DFCN_PROBE; not the product).
    python tools/build_variant.py pnprobe -DFCN_PROBE
    FCN_LIB_NAME=libfcn_hip_pnprobe.so python tools/pn_probe.py [cfg]
Per kernel shape (layer, K = CRED, N = CPREV): workgroups, mean shader-clock cycles of wave 0 per workgroup and their split
over the phases of the K loop: prologue | load issue | wait for loads + operand transform + LDS stores | barrier | LDS reads +
MFMAs | epilogue (ReLU mask + dz stores); the rest is the statistics reduction + atomics."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from frustum_convnet_amd import _native

fwd = "fwd" in sys.argv[1:]            # python tools/pn_probe.py [cfg] fwd: the forward GEMMs (conv2 / conv3) instead
cfg = ([a for a in sys.argv[1:] if a != "fwd"] + ["car"])[0]
serial = os.environ.get("FCN_SERIAL", "0") == "1"
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg)
data = bench.make_data(cfg, 32, bench.CFGS[cfg][3], 1234, dev)
L = _native.lib()
L.fcn_pn_probe_read.restype = ctypes.c_int
L.fcn_pn_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
buf = np.zeros((32768, 8), dtype=np.uint64)
if fwd:
    L.fcn_pn_probe_read_fwd.restype = ctypes.c_int
    L.fcn_pn_probe_read_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for it in range(3):
        L.fcn_pn_probe_read_fwd(buf.ctypes.data, 32768, 1)
        losses, _ = model(data)
        torch.cuda.synchronize()
        n = L.fcn_pn_probe_read_fwd(buf.ctypes.data, 32768, 1)
    rec = buf[:n].astype(np.int64)
    print("records", n, "(forward GEMMs, scales %s)" % ("serialised" if serial else "concurrent"))
    names = ["prologue", "load issue", "wait+transform+lds st", "barriers", "lds rd + mfma", "y stores"]
    tags = rec[:, 0] >> 16
    for t in sorted(set(tags)):
        r = rec[tags == t]
        full = r[(r[:, 0] & 0xffff) >= 64]                     # full tiles (64 or 128 rows)
        tot = full[:, 1].astype(np.float64)
        ph = full[:, 2:8].astype(np.float64)
        print("conv%d K=%4d N=%4d  wg %5d (full %5d)  cycles/wg %8.0f (max %8.0f) = %.1f us @2.4GHz | " % (
            t >> 32, (t >> 16) & 0xffff, t & 0xffff, len(r), len(full), tot.mean(), tot.max(), tot.mean() / 2400.0) +
              "  ".join("%s %4.1f%%" % (nm, 100 * ph[:, i].mean() / tot.mean()) for i, nm in enumerate(names)) +
              "  rest %4.1f%%" % (100 * (1 - ph.sum(1).mean() / tot.mean())))
    sys.exit(0)
for it in range(3):
    losses, _ = model(data)
    torch.cuda.synchronize()
    L.fcn_pn_probe_read(buf.ctypes.data, 32768, 1)          # forward launches none of the probed kernels: reset
    losses["total_loss"].backward()
    torch.cuda.synchronize()
    n = L.fcn_pn_probe_read(buf.ctypes.data, 32768, 1)
rec = buf[:n].astype(np.int64)
print("records", n, "(scales %s)" % ("serialised" if serial else "concurrent"))
names = ["prologue", "load issue", "wait+transform+lds st", "barriers", "lds rd + mfma", "epilogue mask+store"]
tags = rec[:, 0] >> 16
for t in sorted(set(tags)):
    r = rec[tags == t]
    full = r[(r[:, 0] & 0xffff) == 64]                        # full 64-row tiles
    tot = full[:, 1].astype(np.float64)
    ph = full[:, 2:8].astype(np.float64)
    K = (t >> 16) & 0xffff
    lay = t >> 32
    nm_ = list(names)
    if lay >= 10:                                   # weight-gradient kernels: slot 5 = chunk lookup at the loop top
        nm_[5] = "chunk lookup"
    print("%s %d M/K=%4d N=%4d  wg %5d (full %5d)  cycles/wg %8.0f (max %8.0f) = %.1f us @2.4GHz | " % (
        "wgrad" if lay >= 10 else "dgrad", lay % 10, K, t & 0xffff, len(r), len(full), tot.mean(), tot.max(), tot.mean() / 2400.0) +
          "  ".join("%s %4.1f%%" % (nm, 100 * ph[:, i].mean() / tot.mean()) for i, nm in enumerate(nm_)))
