"""At the bench shape: one training step of the car model with the max-pool taken in conv3's epilogue (keys) and with the pooling
pass over y3 (FCN_POOL_KEYS=0) -- compares every scale's arg-max maps and pooled features (python tools/pool_keys_check.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def run(keys):
    os.environ["FCN_POOL_KEYS"] = keys
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, "car")
    data = bench.make_data("car", 32, 1024, 11, dev)
    losses, metrics = model(data)
    losses["total_loss"].backward()
    torch.cuda.synchronize()
    out = []
    for net in model.feat_net.nets:
        wss = [w for lst in net._pool.free.values() for w in lst]
        assert len(wss) == 1, len(wss)
        out.append((wss[0].amax.cpu(), wss[0].woff.cpu()))
    grads = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu()
    return out, float(losses["total_loss"]), grads


a, la, ga = run("1")
b, lb, gb = run("0")
print("loss", la, lb, "grad max|diff|", float((ga - gb).abs().max()), "of", float(gb.abs().max()))
for s, ((ak, _), (ar, _)) in enumerate(zip(a, b)):
    print("scale", s, "amax", tuple(ak.shape), "differ", int((ak != ar).sum()), "minus-one", int((ak < 0).sum()), int((ar < 0).sum()))
