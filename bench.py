#!/usr/bin/env python
"""bench.py -- frustums/sec of the hot path's training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

One step = the reference's train-loop body on one batch (train/train_net_det.py:120-128): forward of
PointNetDet in train mode (grouping, 4 PointNet scales, FCN, heads, loss) + backward (+ gradient all-reduce
over RCCL when N > 1) + Adam update, on a synthetic KITTI-car-shaped batch already resident in HBM
(B = 32 frustums per GPU, N = 1024 points, strides (0.25,0.5,1,2) -> L = (280,140,70,35)); weak scaling.
The step is captured once into a hipGraph and replayed (no host work in the timed region); --eager disables it.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the dominant kernel (conv GEMM, fp32 32x32x2 MFMA) timed live with HIP events
  cpu_baseline -- the CPU oracle (oracle/det_ref.py + oracle/qdp_ref.c) timed on the host cores, N=1 only
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
NSAMPLE = (32, 64, 64, 128)
MLPS = ((64, 64, 128), (64, 64, 128), (128, 128, 256), (256, 256, 512))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="frustums per GPU")
    ap.add_argument("--npoint", type=int, default=1024)
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture")
    ap.add_argument("--no-optim", action="store_true", help="time forward+backward(+all-reduce) only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-batch", type=int, default=32)
    ap.add_argument("--precision", choices=("split", "f32", "bf16"), default=None,
                    help="MFMA operand mode of the GEMM kernels (default: FCN_PRECISION or 'split')")
    return ap.parse_args()


def build_model(device):
    from frustum_convnet_amd.config import cfg, reset_cfg
    from frustum_convnet_amd import det_base, synth
    reset_cfg()   # defaults == cfgs/det_sample.yaml hot-path keys: HEIGHT_HALF/STRIDE (0.25,0.5,1,2), KITTI
    model = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    synth.fill_state_dict(model.state_dict(), seed=7)
    return model.to(device).train()


def roofline_probe(model, data, reps=20):
    """Times ONLY the conv GEMM launches (layer 2 and 3 of each scale) with HIP events on the current stream
    and returns the roofline object of the dominant kernel template (fwd_gemm_kernel, fp32 MFMA)."""
    from frustum_convnet_amd import _native, pointnet_fused as pf
    L = _native.lib()
    dev = data["point_cloud"].device
    xyz = data["point_cloud"][:, :3].contiguous()
    nets = (model.feat_net.pointnet1, model.feat_net.pointnet2, model.feat_net.pointnet3, model.feat_net.pointnet4)
    tot_ms = 0.0
    tot_flops_exec = 0.0
    tot_flops_dense = 0.0
    tot_bytes_alg = 0.0
    launches = 0
    per = []
    for s, net in enumerate(nets):
        ref = data["center_ref%d" % (s + 1)].contiguous()
        params, bufs = net._param_pack()
        cfgt = (float(net.dist), int(net.nsample), True, 1e-5, 0.1)
        with torch.no_grad():
            feat, idx, cnt, ws, desc, keep = pf._forward_impl(net._pool, cfgt, xyz, ref, None, bufs, params, False)
        pstruct = pf._params_struct(keep[0], keep[1], keep[2], [None] * 3, [None] * 3, [None] * 3)
        E = int(ws.woff[:, -1].sum().item())
        B, Lw, K = desc.B, desc.L, desc.K
        C1, C2, C3 = desc.C1, desc.C2, desc.C3
        for layer, cin, cout in ((2, C1, C2), (3, C2, C3)):
            st = _native.current_stream(dev)
            for _ in range(3):
                _native.check(L.fcn_pn_conv_fwd(ctypes.byref(desc), ctypes.byref(pstruct), ctypes.byref(ws.c), layer, 1, st),
                              "fcn_pn_conv_fwd")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.fcn_pn_conv_fwd(ctypes.byref(desc), ctypes.byref(pstruct), ctypes.byref(ws.c), layer, 1, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl_exec = 2.0 * E * cin * cout
            fl_dense = 2.0 * B * Lw * K * cin * cout
            # compulsory bytes of the launch: operand rows once (conv2 reads the 16-B entries, conv3 the previous
            # pre-BN output), result rows once, the weight matrix once
            tot_bytes_alg += E * (16.0 if layer == 2 else 4.0 * cin) + 4.0 * E * cout + 4.0 * cin * cout
            per.append({"scale": s + 1, "layer": layer, "ms": round(ms, 5), "rows": E,
                        "tflops_executed": round(fl_exec / ms / 1e9, 2)})
            tot_ms += ms
            tot_flops_exec += fl_exec
            tot_flops_dense += fl_dense
            launches += 1
        net._pool.release(ws)
    achieved = tot_flops_exec / tot_ms / 1e9          # TFLOP/s over the 8 launches of one step
    return {"bound": "mfma", "kernel": "fwd_gemm_kernel (conv2/conv3 1x1 GEMMs, fp32 v_mfma_f32_32x32x2)",
            "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": _pmc_traffic(),
            "bytes_per_launch_algorithmic": tot_bytes_alg / launches,
            "flops_per_launch_executed": tot_flops_exec / launches,
            "flops_per_launch_dense_equivalent": tot_flops_dense / launches,
            "avg_launch_ms": round(tot_ms / launches, 5), "launches_per_step": launches, "per_launch": per}


def grouping_roofline(model, data, reps=50):
    """The grouping kernel alone (SURVEY section 8d regime i, HBM-bound scan): the four fcn_query_depth_point_f32 launches of
    one step timed with HIP events on the launch stream; algorithmic bytes = z row + centres + int64 idx + cnt."""
    from frustum_convnet_amd.query_depth_point import query_depth_point
    xyz = data["point_cloud"][:, :3].contiguous()
    B, _, N = xyz.shape
    nets = (model.feat_net.pointnet1, model.feat_net.pointnet2, model.feat_net.pointnet3, model.feat_net.pointnet4)
    refs = [data["center_ref%d" % (s + 1)].contiguous() for s in range(4)]
    nbytes = 0.0
    for net, ref in zip(nets, refs):
        Lw = ref.shape[2]
        nbytes += B * (4.0 * N + 4.0 * Lw + 8.0 * Lw * net.nsample + 4.0 * Lw)

    def run():
        for net, ref in zip(nets, refs):
            query_depth_point(net.dist, net.nsample, xyz, ref)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tbps = nbytes / (ms * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": "qdp_kernel x4 strides (eager launches, includes launch gaps)",
            "bytes_per_step_algorithmic": nbytes, "ms_per_step": round(ms, 5), "achieved": round(tbps, 4), "peak": 8.0,
            "unit": "TB/s", "frac": round(tbps / 8.0, 5)}


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 --pmc pass, if one exists."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("bytes_per_launch")
        except Exception:
            return None
    return None


def cpu_baseline(batch, npoint):
    """The CPU oracle (port of the reference dataflow: dense (B,C,L,K) tensors, torch-CPU conv/BN + C grouping)
    timed on this host: forward + backward of one batch."""
    from oracle import det_ref
    from frustum_convnet_amd import synth
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd import det_base
    reset_cfg()
    m = det_base.PointNetDet(3, num_vec=3, num_classes=2)      # only for the state_dict keys/shapes
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    synth.fill_state_dict(sd, seed=7)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    def step(b):
        data = synth.to_torch(synth.make_batch(b, npoint, seed=1234, variant="car", tilt=(0.01, 0.05)))
        t0 = time.perf_counter()
        _, _, losses = det_ref.forward(sd, data, training=True)
        losses["total_loss"].backward()
        return time.perf_counter() - t0

    # The oracle's dense torch-CPU dataflow does not scale with threads (measured on the GPU box's 256-core host: 2.5
    # frustums/s at 128 threads, 4.1 at 64, 5.2 at 32, 6.0 at 16, 6.3 at 8): time it at 16 and 8 threads and report the
    # better one, with the thread count used.
    saved = torch.get_num_threads()
    best = None
    for n in (16, 8):
        torch.set_num_threads(min(n, saved))
        step(2)                            # warm-up (thread pools, oneDNN primitives)
        t = step(batch)
        if best is None or t < best[0]:
            best = (t, torch.get_num_threads())
    torch.set_num_threads(saved)
    t, cores = best
    return {"value": round(batch / t, 3), "unit": "frustums/s", "cores": cores, "kind": "port",
            "sample": "1 train fwd+bwd step of B=%d N=%d (same synthetic car batch shape), fp32, "
                      "oracle/det_ref.py + oracle/qdp_ref.c, %.2f s, best of 16 / 8 torch threads" % (batch, npoint, t)}


def main():
    a = parse()
    from frustum_convnet_amd import dist as fdist, synth, precision as fprec
    if a.precision:
        fprec.set_precision(a.precision)
    # FCN_BENCH_BACKEND=gloo + FCN_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 path with every rank on GPU 0 (a 1-GPU box
    # cannot form an RCCL communicator); the driver's multi-GPU runs leave both unset.
    one_dev = os.environ.get("FCN_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        torch.cuda.set_device(0)
    rank, world, local = fdist.init_from_env(backend=os.environ.get("FCN_BENCH_BACKEND") or None)
    if one_dev:
        local = 0
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from frustum_convnet_amd.train_state import FlatTrainState
    model = build_model(dev)
    if world > 1:
        fdist.broadcast_state(model, 0)
    # reference optimiser: Adam(lr 1e-3, weight_decay 1e-4), train/train_net_det.py:321-339.  Parameters, gradients and
    # moments are flat buffers: the backward kernels write the gradients in place, the exchange is one all-reduce and the
    # step one streaming kernel (capturable: step counter and hyper-parameters live on the device).
    state = FlatTrainState(model, lr=1e-3, weight_decay=1e-4, world=world)
    optim = not a.no_optim
    data = synth.to_torch(synth.make_batch(a.batch, a.npoint, seed=1234 + rank, variant="car", tilt=(0.01, 0.05)), dev)

    def fwd_bwd():
        losses, _ = model(data)
        losses["total_loss"].backward()
        return losses["total_loss"]

    use_graph = not a.eager
    graph = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                     # allocator / workspace / MIOpen find warm-up, outside capture
            loss = fwd_bwd()
            state.allreduce()
            if optim:
                state.adam_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            if world > 1:
                torch.distributed.barrier()          # no collective in flight while the step is being captured
                torch.cuda.synchronize()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
                loss = fwd_bwd()
                if world == 1 and optim:
                    state.adam_step()
        except Exception as e:  # noqa
            if rank == 0:
                print("[bench] hipGraph capture failed (%s: %s); falling back to eager launches" %
                      (type(e).__name__, e), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
            if world > 1:
                state.allreduce()
                if optim:
                    state.adam_step()
        else:
            fwd_bwd()
            state.allreduce()
            if optim:
                state.adam_step()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tt = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    wall = float(tt.item())
    ms_per_step = wall * 1e3 / a.steps
    final_loss = float(loss.item())

    if rank != 0:
        return
    out = {
        "metric": "frustums/sec (train fwd+bwd) KITTI-car B=32 N=1024",
        "value": round(a.batch * world / (ms_per_step / 1e3), 2),
        "unit": "frustums/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"split": "f32", "f32": "f32", "bf16": "bf16"}[fprec.get_precision()],
        "mfma_operands": {"split": "fp16x3 (forward) / bf16x3 (backward) split of fp32 operands, fp32 accumulate",
                          "f32": "fp32 (v_mfma_f32_32x32x2_f32)", "bf16": "bf16 single term, fp32 accumulate"}[fprec.get_precision()],
        "data": "synthetic",
        "config": {"workload": "cfgs/det_sample.yaml KITTI-car, batch=%d/GPU, Npoint=%d, L=(280,140,70,35), "
                               "train fwd+bwd%s%s" % (a.batch, a.npoint, "" if a.no_optim else "+Adam",
                                                      "+RCCL grad all-reduce" if world > 1 else ""),
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                   "launch": "hipGraph replay" if graph is not None else "eager"},
        "gpu_event_ms_per_step": round(e0.elapsed_time(e1) / a.steps, 4),
        "final_loss": round(final_loss, 5),
    }
    if world == 1 and not a.no_roofline:
        try:
            out["roofline"] = roofline_probe(model, data)
        except Exception as e:  # noqa
            out["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["grouping_roofline"] = grouping_roofline(model, data)
        except Exception as e:  # noqa
            out["grouping_roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1:
        # whole-step HBM roofline (BASELINE.json's "HBM roofline %"): fabric bytes of one step from the committed
        # rocprofv3 --pmc passes (profiles/pmc_traffic.json, tools/gpu_traffic.sh) over THIS command, divided by the
        # step time measured now; peak 8 TB/s (MI355X_MICROARCH.md).  null when no PMC summary is committed.
        try:
            sb = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["step"]["bytes_per_step"]
            tbps = sb / (ms_per_step * 1e-3) / 1e12
            out["hbm_roofline"] = {"bound": "hbm", "bytes_per_step": sb, "bytes_per_frustum": round(sb / a.batch),
                                   "achieved": round(tbps, 3), "peak": 8.0, "unit": "TB/s", "frac": round(tbps / 8.0, 4)}
        except Exception:  # noqa
            out["hbm_roofline"] = None
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_batch, a.npoint)
        except Exception as e:  # noqa
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
