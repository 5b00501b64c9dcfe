"""Inference throughput (eval-mode forward through decode, no labels, hipGraph replay) at the bench shape:
python tools/eval_bench.py [--cfg car] [--batch 32]; FCN_POOL_KEYS=0 times the variant whose pooling pass re-reads y3."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="car")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--min-time", type=float, default=1.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = bench.build_model(dev, a.cfg).eval()
data = bench.make_data(a.cfg, a.batch, bench.CFGS[a.cfg][3], 1234, dev)
data = {k: v for k, v in data.items() if k in ("point_cloud", "one_hot", "rot_angle") or k.startswith("center_ref")}
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(3):
        out = model(data)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    out = model(data)
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
n = max(50, int(a.min_time / ((time.perf_counter() - t0) / 50)))
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"metric": "eval_frustums_per_s", "value": round(a.batch / dt, 1), "ms_per_forward": round(dt * 1e3, 4),
                  "cfg": a.cfg, "batch": a.batch, "pool_keys": os.environ.get("FCN_POOL_KEYS", "1")}))
