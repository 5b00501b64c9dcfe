"""FCN variants timing (graph replay): MIOpen default, MIOpen benchmark mode, NLC matmul formulation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from bench import build_model
dev = torch.device("cuda:0")
model = build_model(dev)
B = 32
feats = [torch.randn(B, c, l, device=dev) for c, l in ((131, 280), (131, 140), (259, 70), (515, 35))]

def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def graphed(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model.zero_grad(set_to_none=True); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay

def fcn_fb():
    x = model.conv_net(*feats); (model.cls_out(x).sum() + model.reg_out(x).sum()).backward()

print("MIOpen default      graph %.3f ms" % timed(graphed(fcn_fb)))

# ---- NLC matmul formulation
def cbr(x, seq, k, s, p):
    # x: (B, L, C) -> conv1d as matmul, BN over (B*L), ReLU
    conv, bn = seq[0], seq[1]
    Bb, L, C = x.shape
    if k == 1:
        a = x.reshape(Bb * L, C); Lo = L
        w = conv.weight[:, :, 0].t()
    else:
        xp = F.pad(x, (0, 0, p, p))
        Lo = (L + 2 * p - k) // s + 1
        cols = [xp[:, t:t + s * (Lo - 1) + 1:s, :] for t in range(k)]
        a = torch.cat(cols, 2).reshape(Bb * Lo, k * C)
        w = conv.weight.permute(2, 1, 0).reshape(k * C, -1)
    y = a @ w
    y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, 0.1, 1e-5)
    return torch.relu(y).view(Bb, Lo, -1)

def dbr(x, seq, k):
    conv, bn = seq[0], seq[1]
    Bb, L, C = x.shape
    w = conv.weight.permute(0, 2, 1).reshape(C, -1)          # (Cin, k*Cout)
    y = (x.reshape(Bb * L, C) @ w).view(Bb * L * k, -1)
    y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, 0.1, 1e-5)
    return torch.relu(y).view(Bb, L * k, -1)

featsT = [f.permute(0, 2, 1).contiguous() for f in feats]
def fcn_nlc():
    cn = model.conv_net
    x1, x2, x3, x4 = featsT
    x = cbr(x1, cn.block1_conv1, 3, 1, 1)
    x = cbr(x, cn.block2_conv1, 3, 2, 1); x = cbr(x, cn.block2_conv2, 3, 1, 1)
    xx1 = x = cbr(torch.cat([x, x2], 2), cn.block2_merge, 1, 1, 0)
    x = cbr(x, cn.block3_conv1, 3, 2, 1); x = cbr(x, cn.block3_conv2, 3, 1, 1)
    xx2 = x = cbr(torch.cat([x, x3], 2), cn.block3_merge, 1, 1, 0)
    x = cbr(x, cn.block4_conv1, 3, 2, 1); x = cbr(x, cn.block4_conv2, 3, 1, 1)
    xx3 = cbr(torch.cat([x, x4], 2), cn.block4_merge, 1, 1, 0)
    xx1 = dbr(xx1, cn.block2_deconv, 1); xx2 = dbr(xx2, cn.block3_deconv, 2); xx3 = dbr(xx3, cn.block4_deconv, 4)
    n = xx1.shape[1]
    x = torch.cat([xx1, xx2[:, :n], xx3[:, :n]], 2).reshape(-1, 768)
    wh = torch.cat([model.cls_out.weight[:, :, 0], model.reg_out.weight[:, :, 0]], 0).t()
    out = x @ wh + torch.cat([model.cls_out.bias, model.reg_out.bias])
    return out
def fcn_nlc_fb():
    fcn_nlc().sum().backward()
# correctness vs module path
with torch.no_grad():
    x = model.conv_net(*feats); ref = torch.cat([model.cls_out(x), model.reg_out(x)], 1).permute(0, 2, 1).reshape(-1, 41)
    got = fcn_nlc()
    print("max abs diff nlc vs module:", float((ref - got).abs().max()), "scale", float(ref.abs().max()))
print("NLC matmul eager    %.3f ms" % timed(fcn_nlc_fb))
print("NLC matmul graph    %.3f ms" % timed(graphed(fcn_nlc_fb)))
torch.backends.cudnn.benchmark = True
print("MIOpen benchmark    graph %.3f ms" % timed(graphed(fcn_fb)))
