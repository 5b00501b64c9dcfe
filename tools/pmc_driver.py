"""Small driver for rocprofv3 --pmc passes: a few fwd+bwd of the 4 PointNet scales at the bench shape."""
import sys, os
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
for it in range(3):
    fs = [net.forward_pooled(xyz, refs[s], data["one_hot"]) for s, net in enumerate(nets)]
    sum(f.sum() for f in fs).backward()
torch.cuda.synchronize()
print("done")
