"""When does each scale's PointNet backward chain start and end inside the REPLAYED graph -- without a profiler (the kernel trace
delays cross-stream releases: EXPERIMENTS.md round 4)?  fcn_stamp launches (device wall clock) in front of and behind each
scale's fcn_pn_backward2 call, on the stream autograd runs that node on; times relative to the end of the ConvFeatNet backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from frustum_convnet_amd import _native, pointnet_fused
from frustum_convnet_amd.train_state import FlatTrainState

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
model.defer_metrics_join = True
state = FlatTrainState(model, lr=1e-3, weight_decay=1e-4)
data = bench.make_data(os.environ.get("CFG", "car"), 32, int(os.environ.get("NPOINT", "1024")), 1234, dev)
ns = model.feat_net.num_scales
slots = torch.zeros(4 + 2 * ns, dtype=torch.int64, device=dev)      # 0 step start, 1 loss done, 2 backward done, 3 adam done, 4+2k / 5+2k scale k
L = _native.lib()
Ls = [net_L for net_L in [None] * ns]


def stamp(i):
    _native.check(L.fcn_stamp(slots.data_ptr() + 8 * i, _native.current_stream(dev)), "fcn_stamp")


orig_bwd = pointnet_fused._PointNetPooled.backward
order = {}


def bwd(ctx, dfeat, a, b):
    k = order.setdefault(int(ctx.desc.L), len(order))
    stamp(4 + 2 * k)
    out = orig_bwd(ctx, dfeat, a, b)
    stamp(5 + 2 * k)
    return out


pointnet_fused._PointNetPooled.backward = staticmethod(bwd)
prefetch = os.environ.get("FCN_PREFETCH", "1") != "0"


def step():
    stamp(0)
    if prefetch:
        model.next_batch = data
    losses, _ = model(data)
    stamp(1)
    model.backward(losses["total_loss"])
    stamp(2)
    state.adam_step()
    stamp(3)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
    step()
R = 40
acc = np.zeros(len(slots))
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
for _ in range(R):
    g.replay()
    torch.cuda.synchronize()
    s = slots.cpu().numpy().astype(np.float64) * 0.01          # the SECOND captured step's stamps (they overwrite the first's)
    acc += s - s[0]
acc /= R
print("second step of a replayed two-step graph, us from its start (mean of %d replays; +%d stamp launches):" % (R, 4 + 2 * ns))
print("  loss done %8.1f   backward done %8.1f   adam done %8.1f" % (acc[1], acc[2], acc[3]))
for Lw, k in sorted(order.items(), key=lambda kv: kv[1]):
    print("  scale L=%-4d (backward node %d)  chain starts %8.1f   all launches of the call enqueued-and-reached %8.1f" % (Lw, k, acc[4 + 2 * k], acc[5 + 2 * k]))
