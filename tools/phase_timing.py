"""Phase timing of one training step with HIP events, eager and as hipGraph replays (diagnostic, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from frustum_convnet_amd import synth

dev = torch.device("cuda:0")
model = build_model(dev)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
xyz = data["point_cloud"][:, :3].contiguous()
refs = [data["center_ref%d" % i] for i in (1, 2, 3, 4)]
nets = (model.feat_net.pointnet1, model.feat_net.pointnet2, model.feat_net.pointnet3, model.feat_net.pointnet4)

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps

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

def report(name, fn):
    model.zero_grad(set_to_none=True)
    ge, we = timed(fn)
    try:
        rp = graphed(fn)
        gg, wg = timed(rp)
    except Exception as e:
        gg = wg = float("nan"); print("   graph capture failed:", type(e).__name__, e)
        torch.cuda.synchronize()
    print("%-28s eager gpu %.3f ms wall %.3f | graph gpu %.3f ms" % (name, ge, we, gg))

for s, net in enumerate(nets):
    def fb(net=net, s=s):
        f = net.forward_pooled(xyz, refs[s], data["one_hot"]); f.sum().backward()
    report("scale %d fwd+bwd" % (s + 1), fb)
def pn_all():
    fs = [net.forward_pooled(xyz, refs[s], data["one_hot"]) for s, net in enumerate(nets)]
    sum(f.sum() for f in fs).backward()
report("4 scales fwd+bwd", pn_all)
feats = [net.forward_pooled(xyz, refs[s], data["one_hot"]).detach() for s, net in enumerate(nets)]
def fcn_fb():
    x = model.conv_net(*feats); (model.cls_out(x).sum() + model.reg_out(x).sum()).backward()
report("FCN+heads fwd+bwd", fcn_fb)
def fcn_f():
    with torch.no_grad():
        x = model.conv_net(*feats); model.cls_out(x); model.reg_out(x)
report("FCN+heads fwd", fcn_f)
# loss tail alone: feed fixed logits through the model's tail by stubbing the feature path
x0 = model.conv_net(*feats).detach()
cls0 = model.cls_out(x0).detach().requires_grad_(True)
reg0 = model.reg_out(x0).detach().requires_grad_(True)
def tail_fb():
    m = model
    of, oc, ocl, orr = m.feat_net.forward, m.conv_net.forward, m.cls_out.forward, m.reg_out.forward
    m.feat_net.forward = lambda *a, **k: (None,) * 4
    m.conv_net.forward = lambda *a: x0
    m.cls_out.forward = lambda x: cls0
    m.reg_out.forward = lambda x: reg0
    try:
        l, _ = m(data); l["total_loss"].backward()
    finally:
        m.feat_net.forward, m.conv_net.forward, m.cls_out.forward, m.reg_out.forward = of, oc, ocl, orr
def tail_wrap():
    cls0.grad = None; reg0.grad = None; tail_fb()
report("loss tail fwd+bwd", tail_wrap)
def full():
    l, _ = model(data); l["total_loss"].backward()
report("full step (grads fresh)", full)
params = [p for p in model.parameters()]
opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-4, capturable=True, fused=True)
def full_opt():
    l, _ = model(data); l["total_loss"].backward(); opt.step()
report("full step + fused Adam", full_opt)
