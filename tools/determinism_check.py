import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from helpers import load_golden, golden_inputs, golden_state_dict
from frustum_convnet_amd import synth
from test_gpu_model import _model
g = load_golden("car_b4_n512")
data = synth.to_torch(golden_inputs(g), "cuda")
outs=[]
for rep in range(2):
    m = _model(g); m.train()
    for it in range(3):
        lo,_ = m(data); cls,reg = m.last_logits
        lo["total_loss"].backward()
        gr = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        outs.append((float(lo["total_loss"]), cls.detach().clone(), reg.detach().clone(), gr.clone()))
        for p in m.parameters(): p.grad=None
        # keep BN running stats evolving but weights fixed
ref=outs[0]
for i,o in enumerate(outs):
    print(i, "%.10f"%o[0], "cls maxdiff %.3e reg %.3e grad %.3e"%(float((o[1]-ref[1]).abs().max()), float((o[2]-ref[2]).abs().max()), float((o[3]-ref[3]).abs().max())))
