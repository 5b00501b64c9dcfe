"""Times fcn_convnet_forward alone (events) -- used with FCN_DBG ablation switches."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from frustum_convnet_amd import fcn_fused
dev = torch.device("cuda:0")
model = build_model(dev)
B = 32
feats = [torch.randn(B, l, c, device=dev) for c, l in ((128, 280), (128, 140), (256, 70), (512, 35))]
oh = torch.zeros(B, 3, device=dev); oh[:, 0] = 1
def f():
    with torch.no_grad():
        return fcn_fused.convnet_fused(model._cn_pool, model.conv_net, model.cls_out, model.reg_out, feats, oh)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): f()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    f()
for _ in range(3): graph.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): graph.replay()
e1.record(); torch.cuda.synchronize()
print("FCN_DBG=%s fcn forward %.3f ms" % (os.environ.get("FCN_DBG", "0"), e0.elapsed_time(e1) / 20))
