"""Inference throughput (eval-mode forward through decode, no labels, hipGraph replay) at the bench shape:
python tools/eval_bench.py [--cfg car] [--batch 32] [--precision split]; FCN_POOL_KEYS=0 times the variant whose pooling pass
re-reads y3.  (The same measurement is the `inference` entry of bench.py's `configs` block.)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="car")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--precision", default="split")
ap.add_argument("--min-time", type=float, default=1.0)
a = ap.parse_args()
r = bench.measure_inference(a.cfg, a.batch, torch.device("cuda:0"), a.min_time, a.precision)
r["pool_keys"] = os.environ.get("FCN_POOL_KEYS", "default")
print(json.dumps(r))
