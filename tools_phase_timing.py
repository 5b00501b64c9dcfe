"""Eager-mode phase timing of one training step with HIP events (diagnostic, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench import build_model
from frustum_convnet_amd import synth

dev = torch.device("cuda:0")
model = build_model(dev)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
xyz = data["point_cloud"][:, :3].contiguous()
refs = [data["center_ref%d" % i] for i in (1, 2, 3, 4)]
nets = (model.feat_net.pointnet1, model.feat_net.pointnet2, model.feat_net.pointnet3, model.feat_net.pointnet4)

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps

for s, net in enumerate(nets):
    g, w = timed(lambda: net.forward_pooled(xyz, refs[s], data["one_hot"]))
    print("scale %d fwd        gpu %.3f ms  wall %.3f ms" % (s + 1, g, w))
    def fb():
        f = net.forward_pooled(xyz, refs[s], data["one_hot"]); f.sum().backward()
    g, w = timed(fb)
    print("scale %d fwd+bwd    gpu %.3f ms  wall %.3f ms" % (s + 1, g, w))
feats = [net.forward_pooled(xyz, refs[s], data["one_hot"]).detach() for s, net in enumerate(nets)]
g, w = timed(lambda: model.conv_net(*feats))
print("FCN fwd            gpu %.3f ms  wall %.3f ms" % (g, w))
def fcn_fb():
    x = model.conv_net(*feats); (model.cls_out(x).sum() + model.reg_out(x).sum()).backward()
g, w = timed(fcn_fb)
print("FCN+heads fwd+bwd  gpu %.3f ms  wall %.3f ms" % (g, w))
def full():
    model.zero_grad(set_to_none=False); l, _ = model(data); l["total_loss"].backward()
g, w = timed(full)
print("full step eager    gpu %.3f ms  wall %.3f ms" % (g, w))
def fwd_only():
    with torch.no_grad(): model(data)
g, w = timed(fwd_only)
print("full fwd (no_grad) gpu %.3f ms  wall %.3f ms" % (g, w))
