import sys, time; sys.path.insert(0,'/root/repo')
import torch, bench
for n in (128, 64, 32, 16, 8):
    torch.set_num_threads(n)
    r = bench.cpu_baseline(32, 1024)
    print(n, r["value"], r["sample"][-10:])
