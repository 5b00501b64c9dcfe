import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
data = bench.make_data("car", 32, 1024, 1234, dev)
r = bench.grouping_op_row(data, "car", reps=200)
print(os.environ.get("FCN_LIB_NAME", "prod"), r["ms_per_step"], r["achieved_tbps"], r["frac"])
