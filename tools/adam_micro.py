"""Micro-benchmark of the flat Adam step against plain torch streaming ops of the same size (bandwidth calibration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frustum_convnet_amd.train_state import FlatTrainState
import bench


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


dev = torch.device("cuda", 0)
model = bench.build_model(dev)
st = FlatTrainState(model, lr=1e-3, weight_decay=1e-4)
st.grad.normal_()
n = st.numel
print("numel", n, "MB", n * 4 / 1e6)
t = timeit(st.adam_step)
print("adam_step      %.1f us  -> %.2f TB/s (7 accesses)" % (t, 7 * n * 4 / t / 1e6))
x, y, z = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.empty(n, device=dev)
t = timeit(lambda: x.add_(y))
print("x.add_(y)      %.1f us  -> %.2f TB/s (3 accesses)" % (t, 3 * n * 4 / t / 1e6))
t = timeit(lambda: torch.add(x, y, out=z))
print("add(x,y,out=z) %.1f us  -> %.2f TB/s (3 accesses)" % (t, 3 * n * 4 / t / 1e6))
t = timeit(lambda: z.copy_(x))
print("z.copy_(x)     %.1f us  -> %.2f TB/s (2 accesses)" % (t, 2 * n * 4 / t / 1e6))
big = torch.randn(64 * 1024 * 1024, device=dev)
big2 = torch.empty_like(big)
t = timeit(lambda: big2.copy_(big), 20)
print("256MB copy     %.1f us  -> %.2f TB/s" % (t, 2 * big.numel() * 4 / t / 1e6))
