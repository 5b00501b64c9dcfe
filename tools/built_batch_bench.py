"""The `configs` row "train, batch BUILT on the device every step" of bench.py alone, several times (one line each):
python tools/built_batch_bench.py [repeats] [min_time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sys.argv = [sys.argv[0]] + sys.argv[3:]
rep = int(os.environ.get("REP", "3"))
a = bench.parse()
dev = torch.device("cuda", 0)
for i in range(rep):
    for build in (True, False):
        m = bench.measure(a, "car", "split", 20, 10, float(os.environ.get("MIN_TIME", "0.5")), dev, 0, 1, build_inputs=build)
        print("built" if build else "resident", round(m["ms_per_step"], 4), "ms", round(32e3 / m["ms_per_step"], 1), "frustums/s", "loss", round(m["final_loss"], 5), flush=True)
        del m
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
