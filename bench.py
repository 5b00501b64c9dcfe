#!/usr/bin/env python
"""bench.py -- frustums/sec of the hot path's training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg car|people|refine|sunrgbd] [--precision split|f32|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

One step = the reference's train-loop body on one batch (train/train_net_det.py:120-133): forward of PointNetDet in train
mode (grouping, 4 PointNet scales, FCN, heads, loss + metrics) + backward (+ gradient all-reduce over RCCL when N > 1) +
Adam update, on a synthetic batch already resident in HBM (default: KITTI-car-shaped, B = 32 frustums per GPU, N = 1024
points, strides (0.25,0.5,1,2) -> L = (280,140,70,35)); weak scaling.  The step is captured into hipGraph(s) and replayed
(no host work in the timed region besides the replays and, for N > 1, the collective calls); --eager disables it.

Timing: W warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize; when K steps take less than --min-time
seconds (default 1 s) the K-step window is repeated back to back (`rounds` of K steps each, all inside ONE bracket) and
ms_per_step is the mean over rounds * K steps -- a 40 ms window says little about a GPU that clocks by power budget.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- the top-time kernel family of the step, timed LIVE (HIP events around its C-ABI entry point, on the stream
                  it is launched on, eager launches of the same step), with a `kernels` table for every entry point
  configs      -- short timed windows (>= 0.35 s each) of the other BASELINE.json configurations: people, refine, the
                  SUN-RGBD variant and the bf16 throughput mode of the car config (N=1, default car / split invocation only)
  cpu_baseline -- the CPU oracle (oracle/det_ref.py + oracle/qdp_ref.c) timed on the host cores, N=1 only, <= ~30 s
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# MI355X_MICROARCH.md: dense MFMA peaks.  fp32 MFMA = the fp32 vector rate; 16-bit MFMA 2.5 PFLOP/s dense.
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_16BIT_MFMA_TFLOPS = 2500.0
PEAK_HBM_TBPS = 8.0
NSAMPLE = (32, 64, 64, 128)

CFGS = {
    # name: (yaml, strides, z_range of the synthetic frustums (None: 0..70 m), default N)
    "car": ("cfgs/det_sample.yaml", (0.25, 0.5, 1.0, 2.0), None, 1024),
    "people": ("cfgs/det_sample_people.yaml", (0.1, 0.2, 0.4, 0.8), None, 1024),
    "refine": ("cfgs/refine_car.yaml", (0.1, 0.2, 0.4, 0.8), (-1.0, 1.0), 512),
    # the five-scale SUN-RGBD variant: 8 m of depth, 10 classes, models/det_base_sunrgbd.py
    "sunrgbd": ("cfgs/det_sample_sunrgbd.yaml", (0.1, 0.2, 0.4, 0.8, 1.6), None, 2048),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="frustums per GPU")
    ap.add_argument("--npoint", type=int, default=None)
    ap.add_argument("--cfg", choices=sorted(CFGS), default="car")
    ap.add_argument("--min-time", type=float, default=1.0, help="minimum length of the timed region in seconds")
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture")
    ap.add_argument("--no-optim", action="store_true", help="time forward+backward(+all-reduce) only")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce after the whole backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short windows of the other configurations")
    ap.add_argument("--cpu-baseline-batch", type=int, default=32)
    ap.add_argument("--precision", choices=("split", "f32", "bf16", "bf16ops"), default=None,
                    help="MFMA operand mode of the GEMM kernels (default: FCN_PRECISION or 'split')")
    return ap.parse_args()


def build_model(device, cfg_name="car"):
    from frustum_convnet_amd.config import reset_cfg, merge_cfg_from_file
    from frustum_convnet_amd import det_base, synth
    reset_cfg()
    merge_cfg_from_file(os.path.join(ROOT, CFGS[cfg_name][0]))
    model = _new_model(cfg_name)
    synth.fill_state_dict(model.state_dict(), seed=7)
    return model.to(device).train()


def _new_model(cfg_name):
    from frustum_convnet_amd import det_base, det_base_sunrgbd
    if cfg_name == "sunrgbd":
        return det_base_sunrgbd.PointNetDet(3, num_vec=10, num_classes=2)
    return det_base.PointNetDet(3, num_vec=3, num_classes=2)


def _batch_np(cfg_name, batch, npoint, seed):
    from frustum_convnet_amd import synth
    _, strides, z_range, _ = CFGS[cfg_name]
    if cfg_name == "sunrgbd":
        from frustum_convnet_amd.dataset_info import SUNRGBDCategory
        return synth.make_batch(batch, npoint, strides=strides, max_depth=8.0, seed=seed, variant="car", tilt=(0.01, 0.05),
                                z_range=z_range, num_classes=10, mean_sizes=SUNRGBDCategory.MEAN_SIZE_ARRAY)
    return synth.make_batch(batch, npoint, strides=strides, seed=seed, variant="car", tilt=(0.01, 0.05), z_range=z_range)


def make_data(cfg_name, batch, npoint, seed, device):
    from frustum_convnet_amd import synth
    return synth.to_torch(_batch_np(cfg_name, batch, npoint, seed), device)


# ------------------------------------------------------------------------------------------------
# Live per-entry-point timing: HIP events recorded on the launch stream right around each C-ABI call of an EAGER step.
# Every launch of a call is enqueued back to back on that stream, so the interval is the GPU time of the call's kernels.
TIMED = ("fcn_pn_group_compact2", "fcn_pn_forward", "fcn_convnet_pack", "fcn_convnet_forward2", "fcn_det_loss_tail_rows2",
         "fcn_det_iou_metrics", "fcn_convnet_backward", "fcn_pn_backward2", "fcn_adam_step_f32")


def _entry_key(name, a):
    if name in ("fcn_pn_forward", "fcn_pn_backward2"):
        d = a[0]._obj                      # the PnDesc behind ctypes.byref()
        return "%s[L=%d,K=%d,C=%d-%d-%d]" % (name, d.L, d.K, d.C1, d.C2, d.C3)
    return name


class CallTimer:
    """Brackets every C-ABI call of TIMED on the stream it is launched on: with HIP events (eager launches), or -- slots given --
    with fcn_stamp launches (device wall clock, 100 MHz) that are CAPTURED with the step and fire inside every replay."""

    def __init__(self, lib, slots=None, dev=None):
        self.lib, self.orig, self.rec, self.slots, self.dev = lib, {}, [], slots, dev

    def __enter__(self):
        from frustum_convnet_amd import _native
        for name in TIMED:
            fn = getattr(self.lib, name)
            self.orig[name] = fn

            def wrap(*a, _fn=fn, _name=name):
                if self.slots is not None:
                    k = len(self.rec)
                    if 2 * k + 2 > self.slots.numel():
                        return _fn(*a)
                    stamp = self.orig_stamp
                    _native.check(stamp(self.slots.data_ptr() + 16 * k, _native.current_stream(self.dev)), "fcn_stamp")
                    rc = _fn(*a)
                    _native.check(stamp(self.slots.data_ptr() + 16 * k + 8, _native.current_stream(self.dev)), "fcn_stamp")
                    self.rec.append((_entry_key(_name, a), None, None, None))
                    return rc
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                self.rec.append((_entry_key(_name, a), e0, e1, None))
                return rc
            setattr(self.lib, name, wrap)
        self.orig_stamp = self.lib.fcn_stamp
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.lib, name, fn)


def fcn_flops(B, Ls, nvec=3, c1=128, nout=41):
    """MACs of ConvFeatNet + heads x 2, for 4 levels (c1 = 128, 41 head columns) or 5 (c1 = 64, 69)."""
    w = [c1, 128, 256, 512, 512][:len(Ls)]          # block widths = pooled feature widths from level 2 on
    macs = (128 + nvec) * w[0] * 3 * Ls[0]
    for j in range(1, len(Ls)):
        macs += w[j - 1] * w[j] * 3 * Ls[j] + w[j] * w[j] * 3 * Ls[j] + (2 * w[j] + nvec) * w[j] * Ls[j]
        macs += w[j] * 256 * (1 << (j - 1)) * Ls[j]                      # deconvolution, kernel = stride = 2^(j-1)
    macs += 256 * (len(Ls) - 1) * nout * Ls[1]
    return 2.0 * macs * B


def kernel_table(model, state, data, optim, prec, reps=10, graph=True):
    """Per C-ABI entry point: launches per step, live GPU time, executed FLOPs / algorithmic bytes, fraction of its roof."""
    from frustum_convnet_amd import _native
    lib = _native.lib()
    B = data["point_cloud"].shape[0]
    Ls = [data["center_ref%d" % i].shape[2] for i in range(1, 6) if ("center_ref%d" % i) in data]
    agg = {}
    model.feat_net.drop_prefetch()                       # (a front the last replayed step prefetched: this table times the whole front)
    dev = data["point_cloud"].device

    def one_step():
        losses, _ = model(data)
        model.backward(losses["total_loss"])
        if optim:
            state.adam_step()

    how = None
    if graph:
        # the step captured ONCE with a device-clock stamp in front of and behind every entry point's launches, on the stream they
        # are launched on: the intervals are taken inside the REPLAYED graph, the form the headline times
        try:
            slots = torch.zeros(512, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with CallTimer(lib, slots=slots, dev=dev) as ct:
                with torch.cuda.graph(g):
                    one_step()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            for rep in range(reps):
                g.replay()
                torch.cuda.synchronize()
                sl = slots.cpu().numpy().astype(np.float64) * 1e-5          # 100 MHz ticks -> ms
                for k, (key, _, _, _) in enumerate(ct.rec):
                    r = agg.setdefault(key, {"ms": 0.0, "calls": 0})
                    r["ms"] += sl[2 * k + 1] - sl[2 * k]
                    r["calls"] += 1
            how = ("device-clock stamps (fcn_stamp) captured in front of and behind the entry point's launches, on their launch "
                   "stream, read after each of %d REPLAYS of the captured step (one step per graph, whole front in the step); the "
                   "2 x %d stamp launches stretch a replay by a few percent" % (reps, len(ct.rec)))
            del g
        except Exception as e:  # noqa
            print("[bench] stamped capture failed (%s: %s); timing eager launches instead" % (type(e).__name__, e), file=sys.stderr)
            agg = {}
            model.feat_net.drop_prefetch()
            torch.cuda.synchronize()
    if not agg:
        for rep in range(reps + 1):
            with CallTimer(lib) as ct:
                one_step()
                torch.cuda.synchronize()
            if rep == 0:
                continue                                     # first pass: allocator / event warm-up
            for key, e0, e1, _ in ct.rec:
                r = agg.setdefault(key, {"ms": 0.0, "calls": 0})
                r["ms"] += e0.elapsed_time(e1)
                r["calls"] += 1
        how = "HIP events on the launch stream around the call, eager launches, mean of %d steps" % reps
    # executed rows per scale (live entries) from the workspaces of the last forward
    nets = model.feat_net.nets
    E = {}
    for net, Lw in zip(nets, Ls):
        for lst in net._pool.free.values():
            for ws in lst:
                if ws.key[2] == Lw and ws.key[3] == net.nsample:
                    E[(Lw, net.nsample)] = int(ws.woff[:, -1].sum().item())
    peak_mm = {"split": PEAK_16BIT_MFMA_TFLOPS / 3.0, "f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_16BIT_MFMA_TFLOPS,
               "bf16ops": PEAK_16BIT_MFMA_TFLOPS}[prec]
    rows = []
    ff = fcn_flops(B, Ls, nvec=model.feat_net.num_vec, c1=model.conv_net.WIDTHS[0],
                   nout=2 + model.reg_out.weight.shape[0])
    nparam = state.numel
    N = data["point_cloud"].shape[2]
    for key, r in agg.items():
        ms = r["ms"] / reps
        calls = r["calls"] // reps
        row = {"entry": key, "calls_per_step": calls, "ms_per_step": round(ms, 5)}
        flops = nbytes = None
        if key.startswith("fcn_pn_forward") or key.startswith("fcn_pn_backward2"):
            Lw, K = int(key.split("L=")[1].split(",")[0]), int(key.split("K=")[1].split(",")[0])
            C1, C2, C3 = [int(v) for v in key.split("C=")[1].rstrip("]").split("-")]
            e = E.get((Lw, K), 0)
            fwd = 2.0 * e * (C1 * C2 + C2 * C3)
            flops = fwd if "forward" in key else 2.0 * fwd
            row["rows_executed"] = e
        elif key == "fcn_convnet_forward2":
            flops = ff
        elif key == "fcn_convnet_backward":
            flops = 2.0 * ff
        elif key == "fcn_adam_step_f32":
            nbytes = 7.0 * 4.0 * nparam
        elif key == "fcn_pn_group_compact2":
            # algorithmic bytes of the fused front: z row + centres in, entry rows (16 B + 4 B window id) + offsets + counts out
            nbytes = sum(B * (4.0 * N + 12.0 * Lw) + 20.0 * E.get((Lw, net.nsample), 0) + B * 8.0 * Lw
                         for net, Lw in zip(nets, Ls))
        if flops is not None:
            tf = flops / (ms * 1e-3) / 1e12
            row.update(bound="mfma", flops_executed=flops, achieved_tflops=round(tf, 2), frac=round(tf / peak_mm, 4),
                       frac_of_fp32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
        elif nbytes is not None:
            tb = nbytes / (ms * 1e-3) / 1e12
            row.update(bound="hbm", bytes_algorithmic=nbytes, achieved_tbps=round(tb, 4), frac=round(tb / PEAK_HBM_TBPS, 4))
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows, peak_mm, how


def source_hash():
    """sha256 over the kernel sources: ties committed PMC numbers to the code they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "frustum_convnet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def grouping_op_row(data, cfg_name, reps=20):
    """The standalone operator of the path -- fcn_query_depth_point_f32, the drop-in for query_depth_point_cuda.forward
    (ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-65): the (B, L, K) int64 index + counts of every scale of the batch,
    timed with HIP events on the launch stream.  Algorithmic bytes: z row + window centres in, idx + cnt out."""
    from frustum_convnet_amd.query_depth_point import query_depth_point_multi
    from frustum_convnet_amd.det_base import PointNetFeat
    from frustum_convnet_amd.det_base_sunrgbd import PointNetFeat as PointNetFeat5
    pc = data["point_cloud"][:, :3, :].contiguous()
    B, _, N = pc.shape
    scales = (PointNetFeat5 if cfg_name == "sunrgbd" else PointNetFeat).SCALES
    strides = CFGS[cfg_name][1]
    refs = [data["center_ref%d" % (i + 1)] for i in range(len(scales))]

    dzs, ks = [float(v) for v in strides[:len(scales)]], [int(K) for _, K in scales]
    outs = query_depth_point_multi(dzs, ks, pc, refs)           # (outputs allocated once: the launch is what is timed)

    def run():
        query_depth_point_multi(dzs, ks, pc, refs, out=outs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # replayed from a hipGraph like the step itself (eager, the ctypes call's host time -- ~15 us -- is what an event pair sees)
    INNER = 10
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(INNER):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * INNER)
    nbytes = sum(B * (4.0 * N + 4.0 * r.shape[2] + 8.0 * r.shape[2] * K + 4.0 * r.shape[2]) for (mlp, K), r in zip(scales, refs))
    tb = nbytes / (ms * 1e-3) / 1e12
    return {"entry": "fcn_query_depth_point_multi_f32[%d scales in one launch, API form: int64 idx + cnt]" % len(scales), "calls_per_step": 1,
            "ms_per_step": round(ms, 5), "bound": "hbm", "bytes_algorithmic": nbytes, "bytes_per_frustum": round(nbytes / B),
            "achieved_tbps": round(tb, 4), "frac": round(tb / PEAK_HBM_TBPS, 4),
            "note": "standalone operator (the model itself uses the fused front, fcn_pn_group_compact2, which never writes idx)"}


def pmc_traffic(kind, entry=None, cfg_name="car", prec="split"):
    """HBM bytes from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by tools/pmc_summarize.py from
    tools/gpu_traffic.sh, keyed by configuration) -- only when they were measured on THESE kernel sources and for THIS
    configuration.  kind "step": the whole-step record; kind "entry": {"launches_per_step", "bytes_per_step", "bytes_per_launch"}
    of one C-ABI entry point.  -> (record | None, reason | None)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
    except Exception:  # noqa
        return None, "no profiles/pmc_traffic.json"
    if d.get("source_hash") != source_hash():
        return None, "profiles/pmc_traffic.json was measured on other kernel sources (hash %s, now %s)" % (
            d.get("source_hash"), source_hash())
    c = d.get("configs", {}).get(cfg_name)
    if c is None:
        return None, "profiles/pmc_traffic.json holds no PMC passes of the '%s' configuration (has: %s)" % (
            cfg_name, ", ".join(sorted(d.get("configs", {}))) or "none")
    if c.get("precision", "split") != prec:
        return None, "the PMC passes of the '%s' configuration were taken in the '%s' operand mode, this run is '%s'" % (
            cfg_name, c.get("precision", "split"), prec)
    if kind == "entry":
        e = c.get("entries", {}).get(entry)
        return (e, None) if e else (None, "no PMC record for %s" % entry)
    return c.get(kind), None


def cpu_baseline(batch, npoint, cfg_name):
    """The CPU oracle (port of the reference dataflow: dense (B,C,L,K) tensors, torch-CPU conv/BN + C grouping) timed on this
    host: forward + backward.  Two points: best of 16 / 8 torch threads (one B=32 step each) and 1 core (B=4) -- bounded to ~30 s."""
    from oracle import det_ref
    from frustum_convnet_amd import synth
    from frustum_convnet_amd.config import reset_cfg, merge_cfg_from_file
    reset_cfg()
    merge_cfg_from_file(os.path.join(ROOT, CFGS[cfg_name][0]))
    m = _new_model(cfg_name)      # only for the state_dict keys/shapes
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    synth.fill_state_dict(sd, seed=7)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    _, strides, z_range, _ = CFGS[cfg_name]

    def step(b):
        data = synth.to_torch(_batch_np(cfg_name, b, npoint, 1234))
        t0 = time.perf_counter()
        _, _, losses = det_ref.forward(sd, data, strides, training=True)
        losses["total_loss"].backward()
        return time.perf_counter() - t0

    saved = torch.get_num_threads()
    ncpu = os.cpu_count() or saved
    # BOUNDED: <= ~30 s of CPU work in total (the old three-point sweep took 290 s of a 298 s run, 134 s of it one
    # all-core step).  One B=32 step at the thread count that measured best on these hosts (16; the dense torch-CPU dataflow
    # does not scale further: 5.7 frustums/s at 16 threads, 0.25 at 256), one 1-core step at B=4; 8 threads only while the
    # budget lasts.  Warm-ups (thread pools, oneDNN primitives) at B=2.
    budget_t0 = time.perf_counter()
    spent = lambda: time.perf_counter() - budget_t0
    best = None
    for n, limit in ((16, 1e9), (8, 12.0)):
        if spent() > limit:
            break
        torch.set_num_threads(min(n, ncpu))
        step(2)
        t = step(batch)
        if best is None or t < best[0]:
            best = (t, torch.get_num_threads(), 1)
    variants = {}
    if spent() < 24.0:
        torch.set_num_threads(1)
        b1 = max(1, min(batch, 4))
        variants["1_core"] = {"value": round(b1 / step(b1), 3), "cores": 1, "sample": "1 step of B=%d" % b1}
    torch.set_num_threads(saved)
    t, cores, nst = best
    return {"value": round(batch / t, 3), "unit": "frustums/s", "cores": cores, "kind": "port",
            "sample": "%d train fwd+bwd step of B=%d N=%d (%s cfg, same synthetic batch shape), fp32, "
                      "oracle/det_ref.py + oracle/qdp_ref.c, %.2f s/step, best of 16 / 8 torch threads (the dense torch-CPU "
                      "dataflow does not scale further); bounded to ~30 s in total" % (nst, batch, npoint, cfg_name, t),
            "variants": variants, "host_cpus": ncpu, "seconds_spent": round(spent(), 1)}


def measure(a, cfg_name, prec, steps, warmup, min_time, dev, rank, world, batch=None, npoint=None, force_optim=None,
            build_inputs=False):
    """Builds the model + flat train state for one configuration, captures the step into hipGraph(s), warms up and times
    rounds * steps replays inside ONE barrier + synchronize bracket.  Returns a dict with the timing and the live objects
    (model / state / data) the roofline table needs."""
    from frustum_convnet_amd import dist as fdist, precision as fprec
    from frustum_convnet_amd.train_state import FlatTrainState
    fprec.set_precision(prec)
    batch = batch or a.batch
    npoint = npoint or CFGS[cfg_name][3]
    model = build_model(dev, cfg_name)
    model.defer_metrics_join = os.environ.get("FCN_IOU_JOIN", "late") != "early"     # joined by model.backward()
    if world > 1:
        fdist.broadcast_state(model, 0)
    # reference optimiser: Adam(lr 1e-3, weight_decay 1e-4), train/train_net_det.py:321-339.  Parameters, gradients and
    # moments are flat buffers: the backward kernels write the gradients in place, the exchange is a bucketed all-reduce
    # and the step one streaming kernel (capturable: step counter and hyper-parameters live on the device).
    # FCN_BENCH_COMM=rccl1 (main() formed a ONE-rank RCCL group): the collectives of the N > 1 step are really issued -- communicator,
    # communication stream, capture of the calls into the step's graph -- on a one-GPU box
    one_rank_comm = world == 1 and torch.distributed.is_available() and torch.distributed.is_initialized()
    state = FlatTrainState(model, lr=1e-3, weight_decay=1e-4, world=world, force_comm=one_rank_comm)
    optim = (not a.no_optim) if force_optim is None else bool(force_optim)
    data = make_data(cfg_name, batch, npoint, 1234 + rank, dev)
    # build_inputs: every step's batch dict is BUILT on the device from resident RAW frustum records (fcn_prepare_inputs: resample,
    # rotate to the frustum's centre ray, sliding-frustum centres, labels -- datasets/provider_sample.py:137-262) on the prefetch
    # branch of the step before, into the OTHER of two batch buffers: raw points -> step, no resident batch
    sets = None
    if build_inputs:
        if cfg_name != "car":
            raise SystemExit("build_inputs: the car configuration only")
        from frustum_convnet_amd import inputs as finp, synth
        builder = finp.InputBuilder(npoint, CFGS[cfg_name][1], 70.0, random_flip=True, random_shift=True, device=dev)
        recs = synth.make_records(batch, seed=1234 + rank)
        draws = finp.draw([len(r["points"]) for r in recs], npoint, True, True, np.random.RandomState(1234 + rank))
        t_raw = builder.upload(recs, draws, with_seg=False)
        sets = [builder.launch(t_raw, out=builder.alloc(batch, with_seg=False)) for _ in range(2)]
        data = sets[0]
        torch.cuda.synchronize()
    kstep = [0]
    # FCN_BENCH_SPLIT_STEP=1: the N > 1 form of the step with ONE rank and no process group (the all-reduce calls are no-ops): what
    # a rank's step costs on this box without its collectives, beside the N = 1 line
    rehearse = one_rank_comm or os.environ.get("FCN_BENCH_SPLIT_STEP", "0") == "1"
    overlap = (world > 1 or rehearse) and not a.no_overlap and not a.eager
    # N > 1 step forms (FCN_BENCH_COMM_FORM): "captured" (default) = ONE graph of two whole steps, exactly the N = 1 step, with the
    # all-reduce calls captured INTO it as forked branches (RCCL's stream joins the capture); "host" = round 5's three graphs per step
    # with the collectives issued from the host between the replays
    comm_form = os.environ.get("FCN_BENCH_COMM_FORM", "captured")
    if comm_form not in ("captured", "host"):
        raise SystemExit("FCN_BENCH_COMM_FORM must be 'captured' or 'host'")
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() != "nccl":
        comm_form = "host"          # (a CPU transport -- the gloo rehearsal -- blocks the host inside its collectives: not capturable)
    captured_comm = overlap and comm_form == "captured"
    model.split_backward = overlap

    prefetch = os.environ.get("FCN_PREFETCH", "1") != "0" and (world == 1 or (overlap and use_graph_requested(a)))
    skip_comm = world > 1 and os.environ.get("FCN_SKIP_COMM", "0") == "1"      # rehearsal: the step without its collectives
    steps_per_graph = 1

    # FCN_ADAM_LATE=1: only the PointNet bucket's optimiser step (13 % of the parameters) stays on the chain between the backward and
    # the next forward; the [ConvFeatNet + heads] bucket's runs on the packing branch of that forward, in front of the weight packing
    # -- the first thing that reads its result
    late = (os.environ.get("FCN_ADAM_LATE", "1") == "1" and optim and len(state.buckets) == 2 and
            ((world == 1 and not rehearse) or captured_comm))

    if "FCN_BWD_SHARE" in os.environ:           # A/B: "none", or "0:2+1:3" = scale 0's backward on scale 2's stream, scale 1's on scale 3's
        model.feat_net.bwd_share.clear()
        for pair in [x for x in os.environ["FCN_BWD_SHARE"].split("+") if x and x != "none"]:
            sc, host = (int(v) for v in pair.split(":"))
            model.feat_net.share_backward_stream(sc, host)

    def late_bucket0():
        # (on the packing branch's stream) N > 1: the bucket's all-reduce -- started behind the previous step's FCN backward -- first
        state.wait_allreduce(state.buckets[0][0])
        state.adam_step_bucket(0)

    def opt_step():
        if late:
            state.adam_step_bucket(1)
            # armed only now, BEHIND a backward: a forward that runs before any gradient exists must not step the bucket (ADVICE r5:
            # armed from the start, the first warm-up forward applied an Adam step of zero gradients -- weight decay, counter + 1 --
            # and the two buckets' step counters diverged for the rest of the run)
            model._cn_pool.before_pack = late_bucket0
        else:
            state.adam_step()

    pf_point = os.environ.get("FCN_PF_POINT", "fcn_fwd")
    resident = data
    spg = max(2, 2 * (int(os.environ.get("FCN_STEPS_PER_GRAPH", "2")) // 2))      # whole steps per captured graph (even)

    def fwd_bwd():
        data = sets[kstep[0] % 2] if sets else resident          # this step's batch, and the next one's (the other buffer)
        nxt = sets[(kstep[0] + 1) % 2] if sets else resident
        kstep[0] += 1
        if sets and not prefetch:
            builder.launch(t_raw, out=data)                      # (no prefetch branch: built in front of the forward)
        if prefetch and pf_point == "fcn_fwd":
            model.next_batch = nxt                    # forward() starts the prefetch beside the ConvFeatNet forward
            if sets:
                model.next_batch_build = lambda: builder.launch(t_raw, out=nxt)
        losses, _ = model(data)
        if prefetch and pf_point != "fcn_fwd":
            # the next step's batch (the same resident synthetic batch): its batch-only front -- grouping, entry rows, tile lists,
            # input moments -- runs on a side branch beside this step's backward, as a loader prefetches; the work stays INSIDE the
            # timed step, only off its critical path.  Double-buffered workspaces: captured steps alternate between two graphs.
            model.prefetch(nxt, before=(lambda: builder.launch(t_raw, out=nxt)) if sets else None)
        model.backward(losses["total_loss"])          # == loss.backward(), seeded with a cached unit gradient
        return losses["total_loss"]

    use_graph = not a.eager
    graphs = None
    split_scales = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                     # allocator / workspace warm-up, outside capture
            loss = fwd_bwd()
            state.allreduce()
            if optim:
                opt_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if use_graph:
        try:
            if world > 1:
                torch.distributed.barrier()          # no collective in flight while the step is being captured
                torch.cuda.synchronize()
            mode = "thread_local" if (world > 1 or one_rank_comm) else "global"   # RCCL's watchdog thread may query events meanwhile
            if captured_comm:
                # N > 1, default: the step of the N = 1 graph -- same launches, same branches, two whole steps (even / odd workspace
                # set) per graph -- plus the gradient exchange captured INTO it.  Per step:
                #   head              wait [pointnet] all-reduce of the previous step -> Adam of that bucket -> weight fold -> scales ...
                #   packing branch    wait [fcn+heads] all-reduce of the previous step -> Adam of that bucket -> weight packing
                #   ... forward, loss, heads + ConvFeatNet backward (phase 1 of the split backward)
                #   fork              all-reduce [fcn+heads] (12.2 MB) on RCCL's stream, beside the whole PointNet backward AND the
                #                     next step's PointNet forward (its first reader is the next packing)
                #   PointNet backward, every scale abreast as in the N = 1 step (phase 2)
                #   fork              all-reduce [pointnet] (1.1 MB): the only exposed piece (the next head waits for it)
                # and the capture ends by joining both.  No host call between the steps; a rank's step IS the N = 1 step.
                from frustum_convnet_amd.loss_fused import unit_grad
                # (the warm-up ended with an optimiser step; the graph's steps BEGIN with one: one more backward first -- its forward
                # still runs the late [fcn+heads] update the last warm-up step left pending -- so that every gradient is applied once)
                with torch.cuda.stream(side):
                    fwd_bwd()
                    if not skip_comm:
                        state.allreduce()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if state.comm and not skip_comm:
                    _probe_captured_allreduce(dev, world, mode)

                def cap_step():
                    if optim:
                        if late:
                            state.wait_allreduce(state.buckets[1][0])
                            state.adam_step_bucket(1)            # ([fcn+heads]: late_bucket0 on the packing branch of this forward)
                        else:
                            state.wait_allreduce()
                            state.adam_step()
                    if prefetch:
                        model.next_batch = data
                    losses, _ = model(data)
                    lt = losses["total_loss"]
                    pending = model.take_split()
                    lt.backward(gradient=unit_grad(lt.device))
                    if not skip_comm:
                        state.allreduce_bucket_async(0)
                    pending.backward()                           # (joins the IoU-metrics and prefetch branches)
                    if not skip_comm:
                        state.allreduce_bucket_async(1)
                    return lt

                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=mode):
                    model.feat_net.adopt_prefetch()
                    for _ in range(spg if prefetch else 1):
                        loss = cap_step()
                    state.wait_allreduce()                       # RCCL's stream joins the capture: the next replay's head needs both
                steps_per_graph = spg if prefetch else 1
                graphs = ((g,),)
            elif overlap:
                # N > 1: the step is THREE graphs cut where gradients become final, so that each piece's all-reduce starts on the
                # communication stream while the next graph runs: A = [Adam of the PREVIOUS step's (reduced) gradients, forward, loss,
                # FCN + heads backward] -> all-reduce [FCN + heads] -> B = [PointNet backward of the wide scales] -> all-reduce them ->
                # C = [the narrow scales] -> all-reduce them (~0.1 MB: all that is left exposed).  The optimiser step rides at the head
                # of the next A (behind a stream-side wait for the three all-reduces), not as a launch of its own; with the prefetch the
                # graphs exist twice (even / odd workspace set), as the two steps of the N = 1 graph do -- so a rank's step is the same
                # work in the same order as the N = 1 step, plus two graph launches and the collectives.
                from frustum_convnet_amd.loss_fused import unit_grad
                ns = model.feat_net.num_scales
                wide, narrow = list(range(ns // 2, ns)), list(range(ns // 2))
                if os.environ.get("FCN_BENCH_PIECES", "3") == "2":      # (A/B: the PointNet backward as ONE graph, its 1.1 MB exchanged behind it)
                    wide, narrow = list(range(ns)), []
                nset = 2 if prefetch else 1
                # (the eager warm-up ended with an optimiser step; the graphs' steps BEGIN with one: one more backward first, so
                # that every gradient is applied exactly once)
                with torch.cuda.stream(side):
                    fwd_bwd()
                    if not skip_comm:
                        state.allreduce()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                glist, pool = [], None
                for k in range(nset):
                    gA, gB, gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gA, pool=pool, capture_error_mode=mode):
                        # the front prefetched by the previous graph A of the cycle (for k = 0: by the step in front of the capture,
                        # whose place the last graph A of the cycle takes in every replay) -- into the workspace set this step reads
                        model.feat_net.adopt_prefetch()
                        if optim:
                            state.adam_step()
                        if prefetch:
                            model.next_batch = data
                        losses, _ = model(data)
                        loss = losses["total_loss"]
                        pending = model.take_split()
                        loss.backward(gradient=unit_grad(loss.device))
                        model._join_side()               # (the IoU-metrics and prefetch branches end inside this capture)
                    pool = gA.pool()
                    with torch.cuda.graph(gB, pool=pool, capture_error_mode=mode):
                        pending.backward(scales=wide)
                    if narrow:
                        with torch.cuda.graph(gC, pool=pool, capture_error_mode=mode):
                            pending.backward(scales=narrow)
                    else:
                        gC = None
                    glist.append((gA, gB, gC))
                graphs = tuple(glist)
                split_scales = (wide, narrow)
            else:
                # with the prefetch a step consumes the front its predecessor prepared in the OTHER workspace set: the captured
                # graph holds TWO steps (even / odd) -- one replay = two whole steps, which also halves the graph-to-graph gap
                # per step.  `step()` below replays it on every second call.
                glist = []
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=mode):
                    # (the warm-up's last step prefetched this step's front, as the graph's last step does in every replay)
                    model.feat_net.adopt_prefetch()
                    for _ in range(spg if prefetch else 1):
                        loss = fwd_bwd()
                        if world == 1 and optim:
                            opt_step()
                glist.append(g)
                steps_per_graph = spg if prefetch else 1
                graphs = (tuple(glist),)
        except Exception as e:  # noqa
            if rank == 0:
                print("[bench] hipGraph capture failed (%s: %s); falling back to eager launches" %
                      (type(e).__name__, e), file=sys.stderr)
            graphs = None
            overlap = False
            captured_comm = False
            late = False
            model._cn_pool.before_pack = None
            state._pending = {}
            steps_per_graph = 1
            model.split_backward = False
            model._split = model._pending_split = None          # (a capture that died between take_split() and phase 2)
            model.feat_net.drop_prefetch()                      # (a front prefetched inside the dead capture never ran: dropped, not waited for)
            model._pf_xyz = None
            model.next_batch = None
            torch.cuda.synchronize()

    parity = [0]

    def step():
        if graphs is None:
            fwd_bwd()
            state.allreduce()
            if optim:
                state.adam_step()
        elif captured_comm:
            if parity[0] % steps_per_graph == 0:
                graphs[0][0].replay()                # two whole steps, their collectives inside
            parity[0] += 1
        elif overlap:
            gA, gB, gC = graphs[parity[0] % len(graphs)]
            parity[0] += 1
            state.wait_allreduce()                   # the previous step's three pieces (a stream-side wait): graph A begins with Adam
            gA.replay()
            if not skip_comm:
                state.allreduce_bucket_async(0)      # [FCN + heads]: final after graph A
            gB.replay()
            if not skip_comm:
                state.allreduce_scales_async(split_scales[0])      # the wide PointNet scales: final after graph B
            if gC is not None:
                gC.replay()
                if not skip_comm:
                    state.allreduce_scales_async(split_scales[1])      # the narrow ones
        else:
            if parity[0] % steps_per_graph == 0:
                graphs[0][0].replay()
            parity[0] += 1
            if world > 1:
                state.allreduce()
                if optim:
                    state.adam_step()

    if overlap and graphs is not None and not captured_comm:
        steps_per_graph = len(graphs)                # (whole even / odd cycles)
    even = lambda n: ((n + steps_per_graph - 1) // steps_per_graph) * steps_per_graph       # a replay holds whole steps
    for _ in range(even(warmup)):
        step()
    torch.cuda.synchronize()
    # length of one K-step window -> number of rounds for a timed region of at least min_time seconds
    t0 = time.perf_counter()
    for _ in range(even(steps)):
        step()
    torch.cuda.synchronize()
    probe = time.perf_counter() - t0
    rounds = max(1, int(np.ceil(min_time / max(probe, 1e-6))))
    if world > 1:
        rt = torch.tensor([rounds], device=dev)
        torch.distributed.all_reduce(rt, op=torch.distributed.ReduceOp.MAX)
        rounds = int(rt.item())
        torch.distributed.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(even(rounds * steps)):
        step()
    state.wait_allreduce()                           # (N > 1: the last step's collectives belong to the timed region)
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
    nstep = even(rounds * steps)
    if captured_comm and graphs is not None and optim:    # the last captured step's gradients: its successor's head would have applied them
        model._cn_pool.before_pack = None
        state.adam_step()
        torch.cuda.synchronize()
    elif late and model._cn_pool.before_pack is not None:   # the last step's [ConvFeatNet + heads] update, which the next forward would have run
        model._cn_pool.before_pack = None
        state.adam_step_bucket(0)
        torch.cuda.synchronize()
    model._cn_pool.before_pack = None
    Ls = [data["center_ref%d" % i].shape[2] for i in range(1, 6) if ("center_ref%d" % i) in data]
    comm = None
    if world > 1 or one_rank_comm:
        # what the exchange is made of, measured beside the timed region: ranks the communicator really has (an all-reduce of
        # ones) and the time of each bucket's all-reduce alone (events on the current stream around a blocking call)
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        comm = {"backend": torch.distributed.get_backend(), "ranks": int(round(float(ones.item()))), "buckets": [],
                "skipped_in_timed_region": bool(skip_comm)}
        if comm["ranks"] != world:
            raise SystemExit("the process group has %d ranks, WORLD_SIZE says %d" % (comm["ranks"], world))
        pieces = list(state.buckets)
        comm["form"] = ("captured into the step's hipGraph" if captured_comm else
                        "issued from the host between three graph replays per step" if overlap else "one call after the backward")
        if split_scales is not None:        # the pieces the overlapped step really exchanges
            sr = state.scale_ranges
            pieces = [state.buckets[0]] + [("pointnet scales %s" % "+".join(str(k + 1) for k in ks), sr[ks[0]][0], sr[ks[-1]][1])
                                           for ks in split_scales if ks]
        for name, lo, hi in pieces:
            buf = torch.zeros(hi - lo, device=dev)
            for _ in range(2):
                torch.distributed.all_reduce(buf)
            torch.cuda.synchronize()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            for _ in range(5):
                torch.distributed.all_reduce(buf)
            b1.record()
            torch.cuda.synchronize()
            comm["buckets"].append({"name": name, "mbytes": round(4e-6 * (hi - lo), 3), "allreduce_ms": round(b0.elapsed_time(b1) / 5, 4)})
        torch.distributed.barrier()
    return {"captured_comm": captured_comm, "rehearse": rehearse, "model": model, "state": state, "data": data, "graphs": graphs, "overlap": overlap, "optim": optim, "comm": comm,
            "rounds": rounds, "nstep": nstep, "wall": wall, "ms_per_step": wall * 1e3 / nstep,
            "gpu_event_ms_per_step": e0.elapsed_time(e1) / nstep, "final_loss": float(loss.item()),
            "steps_per_graph": steps_per_graph, "prefetch": prefetch, "late_adam": bool(late),
            "batch": batch, "npoint": npoint, "Ls": Ls}


def use_graph_requested(a):
    return not a.eager


def _probe_captured_allreduce(dev, world, mode):
    """Before the step's collectives are captured: ONE small all-reduce captured into a graph of its own, replayed twice and
    checked (every replay sums `world` ones into a fresh copy).  A stack that cannot capture RCCL calls fails HERE, with a clear
    exception that measure() turns into the eager fallback -- not in the middle of the step's capture."""
    src = torch.ones(1024, device=dev)
    buf = torch.zeros(1024, device=dev)
    torch.distributed.all_reduce(buf)                       # (communicator warm-up outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        buf.copy_(src)
        w = torch.distributed.all_reduce(buf, async_op=True)
        w.wait()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    got = float(buf[0].item())
    if got != float(world) or float(buf.sum().item()) != 1024.0 * world:
        raise RuntimeError("a captured all-reduce of ones over %d ranks replayed to %r" % (world, got))


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): re-executes itself under torch.distributed.run
    with N ranks on this node (one per GPU, rendezvous on 127.0.0.1) and returns the launcher's exit code -- the rank-0 child prints
    the JSON line.  Fails loudly (non-zero) when the box has fewer than N GPUs."""
    import socket
    import subprocess
    one_dev = os.environ.get("FCN_BENCH_ONE_DEVICE", "0") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus and not one_dev:
        print("bench.py: --gpus %d but this box has %d GPU(s): refusing to print a line for fewer ranks than asked for"
              % (a.gpus, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def workload_name(cfg_name, batch, npoint, Ls, optim=True, extra=""):
    return "%s KITTI-%s, batch=%d/GPU, Npoint=%d, L=(%s), train fwd+bwd%s%s" % (
        CFGS[cfg_name][0], cfg_name, batch, npoint, ",".join(str(v) for v in Ls), "+Adam" if optim else "", extra)


# the other BASELINE.json configurations (4: people, 5: refine; 2's bf16 throughput mode) and SURVEY f-4's SUN-RGBD variant,
# each timed in a short window of the same kind after the headline line's measurement (N = 1 only)
# (cfg, precision, Npoint or None = the yaml's): BASELINE.json configs[3] quotes the people config at Npoint = 512, its yaml
# (cfgs/det_sample_people.yaml:28 of the reference) says 1024 -- both are run; car in the exact-fp32 MFMA mode sits beside the
# split-precision headline
OTHER_CONFIGS = (("car", "f32", None), ("people", "split", None), ("people", "split", 512), ("refine", "split", None),
                 ("sunrgbd", "split", None), ("car", "bf16", None), ("car", "bf16ops", None))
DTYPE_LABEL = {"split": "f32 (fp32 operands split into two 16-bit parts, 3 MFMAs per product, fp32 accumulate)", "f32": "f32",
               "bf16": "bf16", "bf16ops": "bf16"}


def measure_inference(cfg_name, batch, dev, min_time=0.35, prec="split"):
    """Eval-mode forward through decode (no labels, no grad) of one configuration, captured into a hipGraph and replayed: the
    inference path of train/test_net_det.py:140-190.  -> dict(value frustums/s, ms_per_forward, ...)."""
    from frustum_convnet_amd import precision as fprec
    fprec.set_precision(prec)
    model = build_model(dev, cfg_name).eval()
    data = make_data(cfg_name, batch, CFGS[cfg_name][3], 1234, dev)
    data = {k: v for k, v in data.items() if k in ("point_cloud", "one_hot", "rot_angle", "ref_center", "rgb_prob") or k.startswith("center_ref")}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):
            model(data)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        model(data)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    n = max(50, int(np.ceil(min_time / ((time.perf_counter() - t0) / 50))))
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"cfg": cfg_name, "precision": prec, "mode": "inference", "dtype": "bf16" if prec.startswith("bf16") else "f32",
            "workload": "%s KITTI-%s, batch=%d/GPU, Npoint=%d, L=(%s), eval forward + decode (no labels)" % (
                CFGS[cfg_name][0], cfg_name, batch, CFGS[cfg_name][3],
                ",".join(str(data["center_ref%d" % i].shape[2]) for i in range(1, 6) if ("center_ref%d" % i) in data)),
            "value": round(batch * n / wall, 2), "unit": "frustums/s", "ms_per_step": round(wall * 1e3 / n, 4),
            "timed_steps": n, "timed_seconds": round(wall, 4)}


def _replay_time(fn, dev, inner=10, min_time=0.2):
    """ms per call of `fn` (launches on the current stream, no host syncs) replayed from a hipGraph holding `inner` calls."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    n = max(20, int(np.ceil(min_time / ((time.perf_counter() - t0) / 20))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * inner), n * inner


def inputs_row(dev, batch, npoint=1024):
    """fcn_prepare_inputs -- the reference's ProviderDataset.__getitem__ + collate (datasets/provider_sample.py:
    137-262,291-327,396-397) as ONE launch over resident raw frustum records -> the batch dict; frustums/s and the fraction of the
    HBM roof its algorithmic bytes reach (raw points gathered through the resample indices in, batch tensors out)."""
    from frustum_convnet_amd import inputs as finp, synth
    from frustum_convnet_amd.config import reset_cfg, merge_cfg_from_file
    reset_cfg()
    merge_cfg_from_file(os.path.join(ROOT, CFGS["car"][0]))
    b = finp.InputBuilder(npoint, CFGS["car"][1], 70.0, random_flip=True, random_shift=True, device=dev)
    recs = synth.make_records(batch, seed=1234)
    draws = finp.draw([len(r["points"]) for r in recs], npoint, True, True, np.random.RandomState(1234))
    t = b.upload(recs, draws, with_seg=True)
    out = b.alloc(batch, with_seg=True)
    ms, n = _replay_time(lambda: b.launch(t, out=out), dev)
    nbytes = float(b.algorithmic_bytes(batch, with_seg=True, pt_stride=t["pt_stride"]))
    tb = nbytes / (ms * 1e-3) / 1e12
    return {"cfg": "car", "mode": "input construction (fcn_prepare_inputs: raw frustum records -> batch dict, one launch)",
            "workload": "cfgs/det_sample.yaml KITTI-car, batch=%d, %d..%d raw points per frustum resampled to Npoint=%d, rotate-to-centre, "
                        "random flip + shift, L=(%s) window centres, cls / seg labels" % (
                            batch, min(len(r["points"]) for r in recs), max(len(r["points"]) for r in recs), npoint,
                            ",".join(str(v) for v in b.L)),
            "value": round(batch / (ms * 1e-3), 1), "unit": "frustums/s", "ms_per_step": round(ms, 5), "timed_steps": n,
            "roofline": {"bound": "hbm", "bytes_algorithmic": nbytes, "bytes_per_frustum": round(nbytes / batch),
                         "achieved": round(tb, 4), "peak": PEAK_HBM_TBPS, "unit": "TB/s", "frac": round(tb / PEAK_HBM_TBPS, 5),
                         "note": "one launch of %d workgroups' worth of work: latency-bound, not bandwidth-bound" % batch}}


def detect_row(dev, batch, frames=8, prec="split"):
    """PointNetDet.detect() -- eval forward, decode into label-format boxes, rotated 3-D NMS per (frame, class) group
    (train/test_net_det.py:126-152,193-293) -- on a batch of `batch` frustums that belong to `frames` frames: frames/s."""
    from frustum_convnet_amd import precision as fprec
    fprec.set_precision(prec)
    model = build_model(dev, "car").eval()
    data = make_data("car", batch, CFGS["car"][3], 1234, dev)
    data = {k: v for k, v in data.items() if k in ("point_cloud", "one_hot") or k.startswith("center_ref")}
    data["rot_angle"] = torch.zeros(batch, 1, device=dev)
    group = (torch.arange(batch, device=dev, dtype=torch.int32) * frames // batch).to(torch.int32).contiguous()

    def run():
        model.detect(data, unit_group=group, num_groups=frames, method="nms")
    how = "hipGraph replay"
    try:
        ms, n = _replay_time(run, dev, inner=2)
    except Exception as e:  # noqa  (a host synchronisation inside detect(): timed eagerly)
        how = "eager (%s)" % type(e).__name__
        torch.cuda.synchronize()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
    return {"cfg": "car", "mode": "detect(): eval forward + decode + rotated 3-D NMS", "precision": prec,
            "workload": "cfgs/det_sample.yaml KITTI-car, %d frustums of %d frames (%d per frame, one class), Npoint=%d, "
                        "TEST.METHOD nms: every foreground window a candidate, greedy rotated-IoU suppression per frame" % (
                            batch, frames, batch // frames, CFGS["car"][3]),
            "value": round(frames / (ms * 1e-3), 1), "unit": "frames/s", "frustums_per_s": round(batch / (ms * 1e-3), 1),
            "ms_per_step": round(ms, 4), "timed_steps": n, "launch": how}


def other_configs(a, dev, min_time=0.3, car_flops=None):
    out = []
    for cfg_name, prec, npt in OTHER_CONFIGS:
        try:
            m = measure(a, cfg_name, prec, 20, 10, min_time, dev, 0, 1, npoint=npt)
            out.append({"cfg": cfg_name, "precision": prec, "dtype": DTYPE_LABEL[prec],
                        "workload": workload_name(cfg_name, m["batch"], m["npoint"], m["Ls"], m["optim"]),
                        "value": round(m["batch"] / (m["ms_per_step"] / 1e3), 2), "unit": "frustums/s",
                        "ms_per_step": round(m["ms_per_step"], 4), "timed_steps": m["nstep"],
                        "timed_seconds": round(m["wall"], 4), "final_loss": round(m["final_loss"], 5)})
            if cfg_name == "car" and prec.startswith("bf16") and car_flops:
                # BASELINE.json configs[1] ("1xMI355X bf16"): one bf16 MFMA per product -- executed FLOPs of the step (entry-space
                # rows, real channels: the headline's table) against the dense bf16 MFMA peak
                tf = car_flops / (m["ms_per_step"] * 1e-3) / 1e12
                out[-1]["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(tf / PEAK_16BIT_MFMA_TFLOPS, 4), "flops_executed_per_step": car_flops,
                                       "note": ("bf16 operands, fp32 storage everywhere" if prec == "bf16ops" else
                                                "BASELINE config 2's bf16 line: bf16 operands, bf16 STORAGE of the PointNet's per-entry tensors "
                                                "(y2 / y3 / dy3 / dz2: the streams that reach HBM), fp32 arenas in the L2-resident FCN "
                                                "(round 6: bf16 arenas there made every FCN launch 22-49 % slower)")}
            del m
        except Exception as e:  # noqa
            out.append({"cfg": cfg_name, "precision": prec, "error": "%s: %s" % (type(e).__name__, e)})
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    try:
        out.append(measure_inference("car", a.batch, dev, min_time))
    except Exception as e:  # noqa
        out.append({"cfg": "car", "mode": "inference", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    # SURVEY 8f rows 1 and 2 (the callers either side of the path), and the training step fed by the on-device builder
    for name, fn in (("input construction", lambda: inputs_row(dev, a.batch)), ("detect", lambda: detect_row(dev, a.batch))):
        try:
            out.append(fn())
        except Exception as e:  # noqa
            out.append({"cfg": "car", "mode": name, "error": "%s: %s" % (type(e).__name__, e)})
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    try:
        from frustum_convnet_amd import precision as fprec
        fprec.set_precision("split")
        m = measure(a, "car", "split", 20, 10, min_time, dev, 0, 1, build_inputs=True)
        out.append({"cfg": "car", "precision": "split", "dtype": DTYPE_LABEL["split"],
                    "mode": "train, batch BUILT on the device every step (raw frustum records -> fcn_prepare_inputs on the prefetch "
                            "branch -> step); the headline's batch is resident",
                    "workload": workload_name("car", m["batch"], m["npoint"], m["Ls"], m["optim"]),
                    "value": round(m["batch"] / (m["ms_per_step"] / 1e3), 2), "unit": "frustums/s",
                    "ms_per_step": round(m["ms_per_step"], 4), "timed_steps": m["nstep"],
                    "timed_seconds": round(m["wall"], 4), "final_loss": round(m["final_loss"], 5)})
        del m
    except Exception as e:  # noqa
        out.append({"cfg": "car", "mode": "train, batch built on the device", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    from frustum_convnet_amd import dist as fdist, precision as fprec
    if a.precision:
        fprec.set_precision(a.precision)
    prec = fprec.get_precision()
    npoint = a.npoint or CFGS[a.cfg][3]
    # FCN_BENCH_BACKEND=gloo + FCN_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 path with every rank on GPU 0 (a 1-GPU box
    # cannot form an RCCL communicator); the driver's multi-GPU runs leave both unset.
    one_dev = os.environ.get("FCN_BENCH_ONE_DEVICE", "0") == "1"
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves (`python bench.py --gpus 8` used to measure ONE GPU and print n_gpus 1)
        sys.exit(self_launch(a))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s: the line would not describe the run" % (a.gpus, os.environ.get("WORLD_SIZE")))
    if torch.cuda.device_count() < a.gpus and not one_dev:
        raise SystemExit("--gpus %d but this box has %d GPU(s)" % (a.gpus, torch.cuda.device_count()))
    if one_dev:
        torch.cuda.set_device(0)
    # FCN_BENCH_COMM=rccl1 (N = 1 only): a ONE-rank RCCL group -- the N > 1 step with its collectives really issued, on one GPU
    rccl1 = os.environ.get("FCN_BENCH_COMM", "") == "rccl1" and a.gpus == 1
    rank, world, local = fdist.init_from_env(backend=("nccl" if rccl1 else os.environ.get("FCN_BENCH_BACKEND") or None),
                                             single_rank_group=rccl1)
    if one_dev:
        local = 0
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    m = measure(a, a.cfg, prec, a.steps, a.warmup, a.min_time, dev, rank, world, npoint=npoint)
    model, state, data, graphs, overlap, optim = m["model"], m["state"], m["data"], m["graphs"], m["overlap"], m["optim"]
    rounds, nstep, wall, ms_per_step, final_loss = m["rounds"], m["nstep"], m["wall"], m["ms_per_step"], m["final_loss"]
    gpu_event_ms = m["gpu_event_ms_per_step"]

    if rank != 0:
        _shutdown_group()
        return
    Ls = m["Ls"]
    car_flops = None
    captured_comm = m["captured_comm"]
    comm_note = ""
    if m.get("comm"):
        cm = m["comm"]
        comm_note = "+%s grad all-reduce%s (%s)" % (
            {"nccl": "RCCL", "gloo": "gloo (CPU transport: a rehearsal of the code path, not of xGMI)"}.get(cm["backend"], cm["backend"]),
            (" in a ONE-rank group (rehearsal of the N > 1 step on one GPU)" if world == 1 else "") +
            (", SKIPPED in the timed region" if cm["skipped_in_timed_region"] else ""),
            "2 pieces captured into the step's graph: [FCN + heads] beside the PointNet backward and the next PointNet forward, "
            "[PointNet] behind the backward" if captured_comm else
            "3 pieces issued between graph replays: [FCN + heads] beside the PointNet backward, the wide scales beside the narrow ones"
            if overlap else "one call after the backward")
    elif m["rehearse"]:
        comm_note = " (the N > 1 step form with one rank, no process group: its collectives are no-ops)"
    if graphs is None:
        launch_note = "eager"
    elif overlap and not captured_comm:
        launch_note = "hipGraph replay x%d per step (Adam at the head of the first), %d workspace sets" % (
            3 if graphs[0][2] is not None else 2, len(graphs))
    else:
        launch_note = "hipGraph replay x%d" % len(graphs) + (
            ", %d steps per replay" % m["steps_per_graph"] if m["steps_per_graph"] > 1 else "") + (
            ", the gradient all-reduces captured inside it (Adam of each bucket at the head of the next step)" if captured_comm else "")
    launch_note += (", next batch's grouping front prefetched beside the backward (double-buffered workspaces)" if m["prefetch"] else "") + (
        ", the [ConvFeatNet + heads] bucket's Adam step on the next forward's weight-packing branch" if m.get("late_adam") else "")
    out = {
        "metric": "frustums/sec (train fwd+bwd) %s B=%d N=%d" % (
            {"car": "KITTI-car", "people": "KITTI-people", "refine": "KITTI-refine", "sunrgbd": "SUN-RGBD"}[a.cfg], a.batch, npoint),
        "value": round(a.batch * world / (ms_per_step / 1e3), 2),
        "unit": "frustums/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "rounds": rounds,
        "timed_steps": nstep, "timed_seconds": round(wall, 4),
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_LABEL[prec],
        "mfma_operands": {"split": "fp16x3 (forward) / bf16x3 (backward) split of fp32 operands, fp32 accumulate: fp32-class",
                          "f32": "fp32 (v_mfma_f32_32x32x2_f32)",
                          "bf16": "bf16 single term, fp32 accumulate; the PointNet's y2/y3/dy3/dz2 stored as bf16 (the FCN's L2-resident arenas stay fp32)",
                          "bf16ops": "bf16 single term, fp32 accumulate, fp32 storage"}[prec],
        "data": "synthetic",
        "config": {"workload": workload_name(a.cfg, a.batch, npoint, Ls, not a.no_optim, comm_note),
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                   "launch": launch_note},
        "gpu_event_ms_per_step": round(gpu_event_ms, 4),
        "final_loss": round(final_loss, 5),
    }
    if m.get("comm"):
        out["comm_ranks"] = m["comm"]["ranks"]
        if m["comm"]["backend"] == "nccl":
            out["rccl_ranks"] = m["comm"]["ranks"]   # (only a communicator that IS RCCL is reported under that name)
        out["comm"] = m["comm"]
    if world == 1 and not a.no_roofline:
        try:
            model.split_backward = False
            rows, peak_mm, how = kernel_table(model, state, data, optim, prec, graph=not a.eager)
            top = next(r for r in rows if "frac" in r)
            rl = {"kernel": top["entry"], "calls_per_step": top["calls_per_step"], "bound": top["bound"],
                  "avg_launch_group_ms": top["ms_per_step"] / max(top["calls_per_step"], 1), "frac": top["frac"]}
            if top["bound"] == "mfma":
                rl.update(achieved=top["achieved_tflops"], peak=round(peak_mm, 1), unit="TFLOP/s",
                          flops_executed_per_step=top["flops_executed"],
                          frac_of_fp32_mfma_peak=top["frac_of_fp32_mfma_peak"],
                          peak_note="16-bit dense MFMA peak 2500 TFLOP/s / 3 instructions per fp32-class product" if prec == "split"
                          else ("dense bf16 MFMA peak" if prec.startswith("bf16") else "fp32 MFMA peak"))
            else:
                rl.update(achieved=top["achieved_tbps"], peak=PEAK_HBM_TBPS, unit="TB/s")
            tr, why = pmc_traffic("entry", top["entry"].split("[")[0], a.cfg, prec)
            rl["traffic"] = tr["bytes_per_launch"] if tr else None
            if tr:
                rl["traffic_note"] = ("HBM bytes per launch of this entry point's kernels (%d launches, %.1f MB per step): "
                                      "1024 * (2 * FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes over the "
                                      "replayed step (profiles/pmc_traffic.json)" % (tr["launches_per_step"],
                                                                                     tr["bytes_per_step"] / 1e6))
            if why:
                rl["traffic_note"] = why
            rl["how"] = "top-time C-ABI entry point of one step: %s; FLOPs = executed (entry-space rows, real channels)" % how
            try:
                rows.append(grouping_op_row(data, a.cfg))
            except Exception as e:  # noqa
                rows.append({"entry": "fcn_query_depth_point_f32", "error": "%s: %s" % (type(e).__name__, e)})
            rl["kernels"] = rows
            out["roofline"] = rl
            car_flops = float(sum(r.get("flops_executed", 0.0) for r in rows)) if a.cfg == "car" else None
        except Exception as e:  # noqa
            out["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1:
        # whole-step HBM roofline (BASELINE.json's "HBM roofline %"): fabric bytes of one step from the committed rocprofv3
        # --pmc passes over THIS command and THIS kernel source, divided by the step time measured now.
        st, why = pmc_traffic("step", cfg_name=a.cfg, prec=prec)
        if st:
            sb = st["bytes_per_step"]
            tbps = sb / (ms_per_step * 1e-3) / 1e12
            out["hbm_roofline"] = {"bound": "hbm", "bytes_per_step": sb, "bytes_per_frustum": round(sb / a.batch),
                                   "achieved": round(tbps, 3), "peak": PEAK_HBM_TBPS, "unit": "TB/s",
                                   "frac": round(tbps / PEAK_HBM_TBPS, 4)}
        else:
            out["hbm_roofline"] = {"traffic": None, "note": why}
    rehearsal = m["rehearse"]
    if world == 1 and not a.no_configs and a.cfg == "car" and prec == "split" and not a.eager and not rehearsal:
        # the other BASELINE.json configurations, driver-visible in the same line: free the headline model first
        del model, state, data, graphs, m
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        # the optimiser-EXCLUSIVE figure SURVEY 8d defines the metric on (forward + loss + backward; `value` above also holds the
        # Adam step, i.e. more work): the same graph-replay measurement without the optimiser launch, on this box
        try:
            mx = measure(a, a.cfg, prec, 20, 10, 0.5, dev, 0, 1, npoint=npoint, force_optim=False)
            out["fwd_bwd_excl_optimizer"] = {"value": round(mx["batch"] / (mx["ms_per_step"] / 1e3), 2), "unit": "frustums/s",
                                             "ms_per_step": round(mx["ms_per_step"], 4), "timed_steps": mx["nstep"],
                                             "note": "forward + loss + backward of the same step, no optimiser launch (SURVEY 8d's timed "
                                                     "region); `value` / `ms_per_step` of this line include the Adam step"}
            del mx
        except Exception as e:  # noqa
            out["fwd_bwd_excl_optimizer"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out["configs"] = other_configs(a, dev, car_flops=car_flops)
        fprec.set_precision(prec)
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_batch, npoint, a.cfg)
        except Exception as e:  # noqa
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    _shutdown_group()
    # RCCL prints its version banner through C stdio (buffered when stdout is a pipe or a file: it would land BEHIND the line at
    # exit): drain that buffer first, so that the JSON line is the last line of stdout
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa
        pass
    print(json.dumps(out), flush=True)


def _shutdown_group():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        try:
            torch.distributed.destroy_process_group()
        except Exception:  # noqa
            pass


if __name__ == "__main__":
    main()
