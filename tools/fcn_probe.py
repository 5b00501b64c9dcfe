"""Intra-kernel phase times of the FCN forward kernels (tuning build libfcn_hip_probe.so, -DFCN_PROBE=1; not the product; with
-DFCN_PROBE=2 the issue phase is split further: the columns then read kernel arguments arrived / row divisions done / first loads
issued / prologue / K loop / groups joined).
    FCN_LIB_NAME=libfcn_hip_probe.so python tools/fcn_probe.py [cfg]
Per layer (Ktot, Cout, Lout): workgroups, and the mean / max over workgroups of the phase durations in us
(entry -> loads issued -> prologue done -> K loop done -> groups joined -> stored -> statistics), plus the span first entry ->
last exit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from frustum_convnet_amd import _native

cfg = sys.argv[1] if len(sys.argv) > 1 else "car"
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg)
data = bench.make_data(cfg, 32, bench.CFGS[cfg][3], 1234, dev)
L = _native.lib()
L.fcn_probe_read.restype = ctypes.c_int
L.fcn_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
buf = np.zeros((65536, 8), dtype=np.uint64)
for it in range(3):
    losses, _ = model(data)
    torch.cuda.synchronize()
    n = L.fcn_probe_read(buf.ctypes.data, 65536, 1)
rec = buf[:n].astype(np.int64)
tags = rec[:, 0]
names = ["issue", "prologue", "kloop", "join", "store", "stats"]
print("records", n)
order = []
for t in tags:
    if t not in order:
        order.append(t)
for t in sorted(set(tags), key=lambda t: rec[tags == t][:, 1].min()):
    r = rec[tags == t]
    st = r[:, 1:8].astype(np.float64) / 100.0          # us (100 MHz clock)
    d = np.diff(st, axis=1)
    has_stats = (r[:, 7] != 0)
    span = (st[has_stats, 6].max() if has_stats.any() else st[:, 5].max()) - st[:, 0].min()
    line = "K=%5d N=%4d Lout=%4d  wg %5d  span %6.1f us | " % (t >> 32, (t >> 16) & 0xffff, t & 0xffff, len(r), span)
    for i, nm in enumerate(names):
        col = d[:, i][r[:, i + 2] != 0] if i >= 0 else d[:, i]
        if len(col):
            line += "%s %.1f/%.1f  " % (nm, col.mean(), col.max())
    line += "| start spread %.1f" % (st[:, 0].max() - st[:, 0].min())
    print(line)
