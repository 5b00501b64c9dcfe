"""What the VENDOR's GEMM kernels (hipBLASLt / rocBLAS Tensile assembly, through torch.mm) take for the layer-3 GEMM of the widest
scale -- the "best structure money can buy" beside tools/micro/pgemm.hip's 128 x 128 / two-barrier numbers (EXPERIMENTS 5.1, 6.3).

The x3 split of an fp32-class product is a plain 16-bit GEMM with three times the reduction length:
    A' = [a_hi | a_lo | a_hi]  (M, 3K),   W' = [w_hi | w_hi | w_lo]  (N, 3K),   Y = A' . W'^T
so the library kernel runs EXACTLY the MFMA work of the product kernels (minus their BatchNorm / ReLU / encode / statistics) on operands
that are already encoded.  Shapes: conv3 forward (M = 36 363, K = 256, N = 512) and its data gradient (K = 512, N = 256), each unsplit
(K) and split (3K), 16-bit and -- where the library offers it -- fp32 output.  python tools/micro/blas_ceiling.py
"""
import sys
import torch

M = 36363
dev = torch.device("cuda:0")
torch.manual_seed(3)


def time_us(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    reps = 20
    try:
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    fn()
        run, per = g.replay, reps
    except Exception as e:                      # a library path that does not capture: time eager launches
        print("   (no graph: %s)" % str(e)[:80])
        run, per = fn, 1
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(1, iters // per)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (n * per))
    return best


def mm(a, wt, out_dtype=None):
    if out_dtype is None:
        return lambda: torch.mm(a, wt)
    return lambda: torch.mm(a, wt, out_dtype=out_dtype)


print("torch", torch.__version__, torch.cuda.get_device_name(0))
for lib in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print("library", lib, "not selectable:", e)
        continue
    print("== preferred BLAS library:", lib, "(cublas = rocBLAS on ROCm)")
    for (K, N, what) in ((256, 512, "conv3 forward"), (512, 256, "conv3 data gradient")):
        for mult in (1, 3):
            for dt in (torch.bfloat16, torch.float16):
                Kp = K * mult
                a = (torch.rand(M, Kp, device=dev) * 2 - 1).to(dt)
                w = (torch.rand(N, Kp, device=dev) * 2 - 1).to(dt)          # (N, K): Y = A . W^T
                wt = w.t()                                                 # NT form (no copy)
                wn = w.t().contiguous()                                    # NN form
                useful = 2.0 * M * K * N
                issued = 2.0 * M * Kp * N
                row = []
                for name, f in (("NT 16-bit out", mm(a, wt)), ("NN 16-bit out", mm(a, wn)),
                                ("NT fp32 out", mm(a, wt, torch.float32)), ("NN fp32 out", mm(a, wn, torch.float32))):
                    try:
                        t = time_us(f)
                        row.append("%s %6.1f us (%4.0f TF/s issued)" % (name, t, issued / t * 1e-6))
                    except Exception as e:
                        row.append("%s n/a (%s)" % (name, str(e).split("\n")[0][:60]))
                print("  %-20s M=%d K=%4d%s N=%d %-8s | %s" % (what, M, K, " x3" if mult == 3 else "   ", N,
                                                                 str(dt).replace("torch.", ""), " | ".join(row)))
                sys.stdout.flush()
# fp32 operands through the library (what the reference's cuDNN / ATen path would run): one line per shape
torch.backends.cuda.preferred_blas_library("hipblaslt")
for (K, N, what) in ((256, 512, "conv3 forward"), (512, 256, "conv3 data gradient")):
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    t = time_us(mm(a, w.t()))
    print("  %-20s fp32 operands (library fp32 GEMM), NT: %6.1f us (%4.0f TF/s)" % (what, t, 2.0 * M * K * N / t * 1e-6))
