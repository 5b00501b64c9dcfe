"""Intra-kernel phase times of the FCN BACKWARD roles (tuning build libfcn_hip_probe3.so, -DFCN_PROBE=3; not the product).
    python tools/build_variant.py probe3 -DFCN_PROBE=3;  FCN_LIB_NAME=libfcn_hip_probe3.so python tools/fcn_probe_bwd.py [cfg]
Per (role, Ktot, Cout, C or Lout): workgroups and the mean / max over workgroups of the phase durations in us --
data-gradient tiles: entry -> first loads issued -> prologue done (chunk table, BN-backward coefficients, barrier) -> K loop done
-> groups summed -> outputs stored -> statistics; weight-gradient workgroups: entry -> first loads -> prologue -> K loop -> partial."""
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
    L.fcn_probe_read(buf.ctypes.data, 65536, 1)            # drop the forward's records
    losses["total_loss"].backward()
    torch.cuda.synchronize()
    n = L.fcn_probe_read(buf.ctypes.data, 65536, 1)
rec = buf[:n]
tags = rec[:, 0]
role = (tags >> np.uint64(60)).astype(np.int64)
print("records", n, "(the table holds 65536: later launches of the backward may be cut off)")
NAMES = {1: ["issue", "prologue", "kloop", "join", "store", "stats"], 2: ["issue", "prologue", "kloop", "write"] if os.environ.get("FCN_PROBE", "3") != "4" else ["issue", "prologue", "kloop", "wait-all", "lds-sum", "write"]}
tot = {}
for t in sorted(set(tags.tolist()), key=lambda t: rec[tags == t][:, 1].min()):
    r = rec[tags == np.uint64(t)].astype(np.int64)
    ro = int(t >> 60)
    if ro not in NAMES:
        continue
    st = r[:, 1:8].astype(np.float64) / 100.0
    d = np.diff(st, axis=1)
    nm = NAMES[ro]
    line = "%s K=%5d N=%4d x=%4d  wg %5d  life %5.1f/%5.1f | " % ("dgrad" if ro == 1 else "wgrad", (t >> 32) & 0xfffffff, (t >> 16) & 0xffff,
                                                              t & 0xffff, len(r), 0, 0)
    last = np.array([row[row > 0].max() for row in st])
    life = last - st[:, 0]
    line = line.replace("  0.0/  0.0", "%5.1f/%5.1f" % (life.mean(), life.max()))
    for i, name in enumerate(nm):
        ok = r[:, i + 2] != 0
        if ok.any():
            line += "%s %.1f/%.1f  " % (name, d[ok, i].mean(), d[ok, i].max())
    line += "| span %.1f" % (last.max() - st[:, 0].min())
    print(line)
