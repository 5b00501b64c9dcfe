"""Gradient norms of every PointNet parameter of one golden fixture against the fp64 oracle's norms, in the modes given."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import load_golden, golden_inputs
from test_gpu_model import _model
from frustum_convnet_amd import synth, precision as fprec
case, modes = sys.argv[1], sys.argv[2:]
g = load_golden(case)
res = {}
for mode in modes:
    data = synth.to_torch(golden_inputs(g), "cuda")
    with fprec.precision(mode):
        m = _model(g); m.train()
        losses, _ = m(data)
        losses["total_loss"].backward()
    res[mode] = {k: float(p.grad.double().norm()) for k, p in m.named_parameters()}
n64 = g["grad_norms64"]
print("%-44s %12s %10s " % ("tensor", "fp64 norm", "ref32") + " ".join("%10s" % m for m in modes))
for i, (nm, ref) in enumerate(zip(g["grad_names"], g["grad_norms"])):
    nm = str(nm)
    if "feat_net" not in nm: continue
    d = max(n64[i], 1e-9)
    print("%-44s %12.5f %10.1e " % (nm, n64[i], abs(ref - n64[i]) / d) + " ".join("%10.1e" % (abs(res[m][nm] - n64[i]) / d) for m in modes))
