"""CPU emulation of the split-precision MFMA modes (csrc/gemm_tile.h): every conv / conv-transpose of the oracle
(oracle/det_ref.py) is replaced by a custom autograd function whose forward, data-gradient and weight-gradient products are
formed from 16-bit operand parts (fp16 or bf16; 1, 3 or 6 terms) with exact accumulation, and the logits / parameter
gradients are compared with an fp64 evaluation.  This is what decided fp16x3 for the forward GEMMs (logits |err| 1.3e-5;
bf16x3 misses the 1e-4 bar at 3e-4) and bf16x3 for the backward GEMMs (gradients stay on the fp32 noise floor).
Usage: python tools/split_emulation.py     (CPU only, a few minutes)"""
import sys, numpy as np, torch, torch.nn.functional as F, types
import torch.nn.grad as G
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import det_ref
from helpers import load_golden, golden_inputs, golden_state_dict
from frustum_convnet_amd import synth
CFG={'fwd':None,'bwd':None}   # each: (dtype, nterms) or None
def parts(x,dt,n):
    out=[];r=x
    for i in range(n):
        h=r.to(dt).to(torch.float32); out.append(h); r=r-h
    return out
TERMS={1:[(0,0)],3:[(0,0),(0,1),(1,0)],6:[(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]}
def smm(fn,a,b,cfg):
    """fn(a,b) bilinear; emulate split product with exact (fp64) accumulation rounded to fp32"""
    if cfg is None or a.dtype==torch.float64: return fn(a,b)
    dt,n=cfg; ns={1:1,3:2,6:3}[n]
    ap=parts(a,dt,ns); bp=parts(b,dt,ns)
    acc=None
    for i,j in reversed(TERMS[n]):
        t=fn(ap[i].double(),bp[j].double())
        acc=t if acc is None else acc+t
    return acc.float()
class Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx,x,w,kind,stride,pad):
        ctx.save_for_backward(x,w); ctx.k=(kind,stride,pad)
        if kind=='c2': f=lambda a,b:F.conv2d(a,b)
        elif kind=='c1': f=lambda a,b:F.conv1d(a,b,stride=stride,padding=pad)
        else: f=lambda a,b:F.conv_transpose1d(a,b,stride=stride)
        return smm(f,x,w,CFG['fwd'])
    @staticmethod
    def backward(ctx,dy):
        x,w=ctx.saved_tensors; kind,stride,pad=ctx.k
        c=CFG['bwd']
        if kind=='c2':
            dx=smm(lambda a,b:G.conv2d_input(x.shape,b,a),dy,w,c)
            dw=smm(lambda a,b:G.conv2d_weight(a,w.shape,b),x,dy,c)
        elif kind=='c1':
            dx=smm(lambda a,b:G.conv1d_input(x.shape,b,a,stride=stride,padding=pad),dy,w,c)
            dw=smm(lambda a,b:G.conv1d_weight(a,w.shape,b,stride=stride,padding=pad),x,dy,c)
        else:
            dx=smm(lambda a,b:F.conv1d(a,b,stride=stride),dy,w,c)
            dw=smm(lambda a,b:G.conv1d_weight(a,w.shape,b,stride=stride),dy,x,c)
        return dx,dw,None,None,None
ns=types.SimpleNamespace(**{k:getattr(F,k) for k in dir(F) if not k.startswith('__')})
ns.conv2d=lambda x,w: Conv.apply(x,w,'c2',1,0)
def c1(x,w,b=None,stride=1,padding=0):
    y=Conv.apply(x,w,'c1',stride,padding); return y if b is None else y+b.view(1,-1,1)
ns.conv1d=c1
ns.conv_transpose1d=lambda x,w,stride=1: Conv.apply(x,w,'ct',stride,0)
det_ref.F=ns
def run(sd,data,strides):
    sd={k:v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k,v in sd.items()}
    _,_,losses=det_ref.forward(sd,data,strides,training=True)
    losses['total_loss'].backward()
    return {k:v.grad for k,v in sd.items() if v.grad is not None}, float(losses['total_loss'])
for case in ["car_b4_n512","refine_b4_n512"]:
    g=load_golden(case); data=synth.to_torch(golden_inputs(g)); sd=golden_state_dict(g)
    strides=tuple(g["meta_strides"])
    sd64={k:(v.double() if v.dtype.is_floating_point else v) for k,v in sd.items()}
    d64={k:(v.double() if v.dtype.is_floating_point else v) for k,v in data.items()}
    CFG['fwd']=CFG['bwd']=None
    g64,l64=run(sd64,d64,strides)
    for name,fw,bw in [("fp32",None,None),("f16x3/bf16x3",(torch.float16,3),(torch.bfloat16,3)),
                       ("f16x3/bf16x6",(torch.float16,3),(torch.bfloat16,6)),
                       ("f16x3/f16x3",(torch.float16,3),(torch.float16,3)),
                       ("bf16x1/bf16x1",(torch.bfloat16,1),(torch.bfloat16,1))]:
        CFG['fwd']=fw; CFG['bwd']=bw
        gg,l=run(sd,data,strides)
        worst_el=0;worst_nm=0;wk=None
        for k in gg:
            r=g64[k]; e=(gg[k].double()-r).abs().max().item()/(r.abs().max().item()+1e-30)
            en=(gg[k].double()-r).norm().item()/(r.norm().item()+1e-30)
            if e>worst_el: worst_el=e;wk=k
            worst_nm=max(worst_nm,en)
        print(case,name,"loss relerr %.2e"%(abs(l-l64)/l64),"worst elem err/max %.2e (%s)  worst norm relerr %.2e"%(worst_el,wk,worst_nm))
