"""Phase boundaries of one step INSIDE the replayed hipGraph, without a profiler: fcn_stamp launches (device wall clock)
at the joins of the captured step.  Backward boundaries come from tensor hooks (they run on the stream of the node that
produced the gradient)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from frustum_convnet_amd import synth, _native, fcn_fused
from frustum_convnet_amd.train_state import FlatTrainState

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
state = FlatTrainState(model, lr=1e-3, weight_decay=1e-4)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
NAMES = ["start", "pointnet_fwd_done", "fcn_fwd_done", "loss_done", "loss_bwd_done", "fcn_bwd_done", "backward_done", "adam_done"]
slots = torch.zeros(len(NAMES), dtype=torch.int64, device=dev)
L = _native.lib()


def stamp(i):
    _native.check(L.fcn_stamp(slots.data_ptr() + 8 * i, _native.current_stream(dev)), "fcn_stamp")


# wrap the phases of PointNetDet.forward
orig_feat = model.feat_net.forward
def feat_forward(*a, **k):
    out = orig_feat(*a, **k)
    stamp(1)
    for t in out:
        if t.requires_grad:
            t.register_hook(lambda g: None)
    out[3].register_hook(lambda g: (stamp(5), None)[1]) if out[3].requires_grad else None
    return out
model.feat_net.forward = feat_forward
orig_cf = fcn_fused.convnet_fused
def cf(*a, **k):
    out = orig_cf(*a, **k)
    stamp(2)
    if out.requires_grad:
        out.register_hook(lambda g: (stamp(4), None)[1])
    return out
fcn_fused.convnet_fused = cf


def step():
    stamp(0)
    losses, _ = model(data)
    stamp(3)
    losses["total_loss"].backward()
    stamp(6)
    state.adam_step()
    stamp(7)


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
acc = np.zeros(len(NAMES) - 1)
tot = []
R = 30
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
for _ in range(R):
    g.replay()
    torch.cuda.synchronize()
    s = slots.cpu().numpy().astype(np.float64) * 0.01       # 100 MHz ticks -> us
    acc += np.diff(s)
    tot.append(s[-1] - s[0])
print("phase (us, mean of %d isolated replays; stamps add ~8 launches):" % R)
for n, v in zip(NAMES[1:], acc / R):
    print("  -> %-20s %8.1f" % (n, v))
print("  total %.1f us" % (sum(tot) / R))
