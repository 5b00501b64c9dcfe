"""Driver for the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: only the roofline probe's conv GEMM launches
(fcn_pn_conv_fwd: conv2 and conv3 of the four scales at the bench shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from frustum_convnet_amd import synth
dev = torch.device("cuda:0")
model = bench.build_model(dev)
data = synth.to_torch(synth.make_batch(32, 1024, seed=1234, variant="car", tilt=(0.01, 0.05)), dev)
r = bench.roofline_probe(model, data, reps=4)
print("probe", r["achieved"], r["avg_launch_ms"])
