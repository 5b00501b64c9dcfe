"""Gradient norms of one golden fixture in a given operand mode against the fp64 oracle's norms (fixtures of make_golden.py full)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import load_golden, golden_inputs
from test_gpu_model import _model
from frustum_convnet_amd import synth, precision as fprec
case, mode = sys.argv[1], sys.argv[2]
g = load_golden(case)
data = synth.to_torch(golden_inputs(g), "cuda")
with fprec.precision(mode):
    m = _model(g); m.train()
    losses, _ = m(data)
    losses["total_loss"].backward()
named = dict(m.named_parameters())
n64 = g["grad_norms64"]
for i, (nm, ref) in enumerate(zip(g["grad_names"], g["grad_norms"])):
    nm = str(nm)
    if "feat_net" in nm and (".conv1." in nm or ".conv2.0" in nm):
        got = float(named[nm].grad.double().norm())
        print("%-6s %-40s got %.6f ref32 %.6f fp64 %.6f  rel(got-64) %.2e rel(ref32-64) %.2e" % (mode, nm, got, ref, n64[i], abs(got - n64[i]) / max(n64[i], 1e-9), abs(ref - n64[i]) / max(n64[i], 1e-9)))
