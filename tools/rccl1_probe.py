"""One-rank RCCL probe: communicator set-up and an all-reduce on ONE MI355X, eager and captured into a hipGraph,
step by step with faulthandler on -- which step a stack survives is the finding."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def say(*a):
    print(*a, flush=True)


torch.cuda.set_device(0)
say("torch", torch.__version__, "hip", torch.version.hip, "nccl", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None)
kw = {}
if os.environ.get("PROBE_DEVICE_ID", "0") == "1":
    kw["device_id"] = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29531", **kw)
say("group up:", dist.get_backend(), dist.get_world_size())
x = torch.ones(1024, device="cuda")
say("eager all_reduce ...")
dist.all_reduce(x)
torch.cuda.synchronize()
say("  ->", float(x[0]), "librccl mapped:", "librccl" in open("/proc/self/maps").read())
y = torch.ones(4 << 20, device="cuda")
w = dist.all_reduce(y[: 1 << 20], async_op=True)
w.wait()
torch.cuda.synchronize()
say("async slice all_reduce ok", float(y[0]))
dist.broadcast(x, src=0)
torch.cuda.synchronize()
say("broadcast ok")
if os.environ.get("PROBE_CAPTURE", "1") == "1":
    g = torch.cuda.CUDAGraph()
    src = torch.ones(1024, device="cuda")
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.copy_(src)
        w = dist.all_reduce(x, async_op=True)
        w.wait()
        x.mul_(2.0)
    say("captured")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    say("replayed ->", float(x[0]))
dist.destroy_process_group()
say("done")
