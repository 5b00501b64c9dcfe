"""Flat training state of the step loop (reference: train/train_net_det.py:114-137 -- forward, loss.mean(),
backward, optimizer.step() -- with optim.Adam from :321-339 and nn.DataParallel from :308-309).

Parameters, gradients and both Adam moments are four contiguous fp32 buffers (13.3 MB each for PointNetDet):
  * every nn.Parameter is re-homed as a view of `flat`, its .grad is a permanent view of `grad`;
  * the HIP backward kernels write each weight gradient STRAIGHT into its view (no autograd accumulation kernels,
    no per-tensor allocations) -- gradients are therefore overwritten, not accumulated, by every backward;
  * cls_out / reg_out weights (and biases) are adjacent, so the fused heads GEMM reads them as one matrix without a cat;
  * the data-parallel exchange is an RCCL all-reduce (sum) of `grad` in TWO buckets cut where the backward finishes them:
    [ConvFeatNet + heads] (2.9 M of the 3.3 M parameters) is final when the FCN backward ends and is reduced on RCCL's
    stream while the PointNet backward still runs; [PointNet] follows it.  The 1/world of the mean is folded into the
    optimiser kernel's grad_scale;
  * the optimiser step is a streaming HIP kernel PER BUCKET (fcn_adam_step_f32, or fcn_sgd_step_f32 for the reference's 'sgd'
    branch) whose step counters and hyper-parameters live in device memory, so it can be captured into the step's hipGraph and
    the learning rate changed between replays (set_lr + lr_for_epoch = the reference's StepLR / MultiStepLR + MIN_LR clamp).
    adam_step_bucket(i) steps one bucket alone; bench.py steps BOTH buckets after the whole backward (adam_step()): stepping the
    [FCN + heads] bucket early, beside the PointNet backward, measured 2 % slower -- the per-bucket form
    stays in the API and is covered by tests/test_gpu_train_state.py.
"""
import ctypes
import weakref

import torch
import torch.distributed as dist

from . import _native


def _ordered_parameters(model):
    """model.parameters() with the two head weights adjacent and the two head biases adjacent (when present)."""
    named = list(model.named_parameters())
    heads = [n for n, _ in named if n in ("cls_out.weight", "reg_out.weight", "cls_out.bias", "reg_out.bias")]
    rest = [(n, p) for n, p in named if n not in heads]
    d = dict(named)
    tail = [(n, d[n]) for n in ("cls_out.weight", "reg_out.weight", "cls_out.bias", "reg_out.bias") if n in d]
    return rest + tail


def lr_for_epoch(epoch, base_lr, lr_steps, gamma, min_lr=0.0):
    """Learning rate of the reference's schedule at `epoch` (train/train_net_det.py:98-105,334-339): MultiStepLR over
    `lr_steps` when there are several milestones, StepLR(step_size = lr_steps[0]) when there is one, clamped from below by
    cfg.TRAIN.MIN_LR.  Feed it to FlatTrainState.set_lr() at the top of every epoch."""
    if len(lr_steps) > 1:
        lr = base_lr * gamma ** sum(1 for m in lr_steps if epoch >= m)
    else:
        lr = base_lr * gamma ** (epoch // int(lr_steps[0]))
    return max(lr, min_lr) if min_lr > 0 else lr


STEP_SPAN = 2048        # elements per workgroup of adam_kernel (ADAM_T * ADAM_V * 4, csrc/optim.hip)


class FlatTrainState:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world=1, group=None,
                 optimizer="adam", momentum=0.9, force_comm=False):
        if optimizer not in ("adam", "sgd"):
            raise ValueError("optimizer must be 'adam' or 'sgd' (cfg.TRAIN.OPTIMIZER, train/train_net_det.py:321-329)")
        self.optimizer = optimizer
        named = _ordered_parameters(model)
        assert named, "model has no parameters"
        self._model = weakref.ref(model)
        # position of every flat-order parameter in model.parameters() order: that is the index torch.optim.Adam (and a
        # checkpoint of the reference, train/train_net_det.py:353,387) uses for its per-parameter state
        model_order = {n: i for i, (n, _) in enumerate(model.named_parameters())}
        self.param_index = [model_order[n] for n, _ in named]
        params = [p for _, p in named]
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatTrainState: fp32 parameters only")
        # every tensor starts on a 16-byte boundary (float4 access in the kernels, heads adjacency is preserved
        # because the head tensors' sizes are handled as one block below)
        # ... and the first tensor after the PointNet scales starts on a multiple of the optimiser kernel's workgroup span
        # (STEP_SPAN elements): the [pointnet] bucket's workgroups and the [fcn+heads] bucket's then tile the buffer exactly like
        # the workgroups of ONE launch over the whole buffer, so both forms share their per-workgroup step counters
        # ... and so does every PointNet scale (feat_net.pointnet<k>.*): adam_step_scale(k) steps ONE scale with the workgroups --
        # and the step-counter slots -- the whole-buffer launch would use for it (the padding holds zeros and stays zero)
        scale_of = lambda n: n.split(".")[1] if n.startswith("feat_net.pointnet") else None
        offs, off, cut_at = [], 0, None
        for k, (n, p) in enumerate(named):
            if k > 0 and named[k - 1][0].startswith("feat_net.") and not n.startswith("feat_net."):
                off = (off + STEP_SPAN - 1) // STEP_SPAN * STEP_SPAN
                cut_at = off
            elif k > 0 and scale_of(n) is not None and scale_of(n) != scale_of(named[k - 1][0]):
                off = (off + STEP_SPAN - 1) // STEP_SPAN * STEP_SPAN
            offs.append(off)
            off += p.numel()
            if not (n in ("cls_out.weight", "cls_out.bias")):
                off = (off + 3) // 4 * 4
        total = (off + 3) // 4 * 4
        self.names = [n for n, _ in named]
        self.params, self.offsets, self.numel = params, offs, total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                gv = self.grad[o:o + n].view(p.shape)
                p.grad = gv
                p._fcn_grad = gv            # the HIP backward writes here and hands autograd no gradient
        self.world, self.group = int(world), group
        # force_comm: issue the collectives even in a world of ONE rank (a 1-rank RCCL communicator on a one-GPU box exercises
        # communicator set-up, the communication stream and hipGraph capture of the calls: tests/test_gpu_dist.py, bench.py's
        # FCN_BENCH_COMM=rccl1 rehearsal); otherwise a world of one never touches torch.distributed
        self.comm = self.world > 1 or bool(force_comm)
        # buckets in the order the backward completes them: everything after the PointNet scales (named feat_net.*,
        # first in the buffer), then the PointNet scales
        cut = cut_at or 0
        self.buckets = [("fcn+heads", cut, total), ("pointnet", 0, cut)] if 0 < cut < total else [("all", 0, total)]
        # the PointNet bucket scale by scale (feat_net.pointnet<k>.*: contiguous, in scale order): the wide scales finish their
        # backward first, so a step loop that differentiates them first (take_split().backward(scales=...)) can start THEIR
        # all-reduce while the narrow scales still run -- only the narrow scales' ~0.1 MB is then exposed behind the backward
        self.scale_ranges = {}
        for n, p, o in zip(self.names, params, offs):
            if n.startswith("feat_net.pointnet"):
                k = int(n[len("feat_net.pointnet"):].split(".")[0]) - 1
                lo, hi = self.scale_ranges.get(k, (o, o))
                self.scale_ranges[k] = (min(lo, o), max(hi, (o + p.numel() + 3) // 4 * 4))
        self._pending = {}          # name -> the torch.distributed work of an all-reduce started and not yet waited for
        if optimizer == "sgd":          # lr, momentum, weight_decay, grad_scale; the momentum buffer lives in exp_avg
            self.hyper = torch.tensor([lr, momentum, weight_decay, 1.0 / self.world, 0.0, 1.0 / self.world], device=dev,
                                      dtype=torch.float32)
        else:
            self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, 1.0 / self.world], device=dev,
                                      dtype=torch.float32)
        # one step counter per workgroup of the optimiser kernel (all equal), indexed by the workgroup's position in the WHOLE
        # buffer: a bucket's launch uses the slots from lo / STEP_SPAN on; step_count is slot 0
        nslots = max(int(_native.lib().fcn_adam_step_slots(ctypes.c_int64(total))), 1)
        assert all(lo % STEP_SPAN == 0 for _, lo, _ in self.buckets)
        self._slot_off = [lo // STEP_SPAN for _, lo, _ in self.buckets]
        self._step_slots = torch.zeros(nslots, device=dev, dtype=torch.int64)
        self.step_count = self._step_slots[0:1]
        self.device = dev

    def set_lr(self, lr):
        """Device-side update: takes effect on the next (eager or replayed) step."""
        self.hyper[0:1].fill_(float(lr))

    def zero_grad(self):
        self.grad.zero_()

    def allreduce(self):
        """Sum of the flat gradient over ranks (the mean's 1/world lives in hyper[5]).  world 1: no-op."""
        if self.comm:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)

    def _start(self, name, lo, hi):
        if name in self._pending:
            raise RuntimeError("FlatTrainState: the all-reduce of '%s' was started twice without a wait_allreduce() in between" % name)
        self._pending[name] = dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def allreduce_bucket_async(self, i):
        """Starts the summing all-reduce of bucket i on the communication stream (it waits for the work enqueued on the
        current stream so far, nothing later): call it right after the backward phase that completes the bucket, keep
        launching the next phase, and call wait_allreduce() before the optimiser step.  Inside a hipGraph capture the call
        becomes a forked branch of the graph (RCCL's stream joins the capture) and wait_allreduce() its join -- every started
        piece must be waited for before the capture ends.  world 1: no-op."""
        if self.comm:
            name, lo, hi = self.buckets[i]
            self._start(name, lo, hi)

    def allreduce_scales_async(self, scales):
        """Starts the summing all-reduce of the gradients of the PointNet scales `scales` (0-based, a contiguous run of scales: one
        call over [first.lo, last.hi)) -- a sub-range of the [pointnet] bucket; every scale must be reduced exactly once per step,
        either through here or through its bucket.  world 1: no-op."""
        ks = sorted(scales)
        assert ks == list(range(ks[0], ks[-1] + 1)) and all(k in self.scale_ranges for k in ks), ks
        lo, hi = self.scale_ranges[ks[0]][0], self.scale_ranges[ks[-1]][1]
        if self.comm:
            self._start("scales %d-%d" % (ks[0], ks[-1]), lo, hi)
        return lo, hi

    def wait_allreduce(self, name=None):
        """The current stream waits for every piece started with allreduce_*_async -- or, with `name` (a bucket's name), for that
        piece alone (no host block on CUDA/HIP; a piece that was never started, or was waited for already, is skipped)."""
        if name is not None:
            w = self._pending.pop(name, None)
            if w is not None:
                w.wait()
            return
        for w in self._pending.values():
            w.wait()
        self._pending = {}

    def adam_step_bucket(self, i):
        """The optimiser step of bucket i alone (its gradients must be final -- and reduced for world > 1 -- on the current
        stream).  Every bucket must be stepped exactly once per training step, in any order, on any streams."""
        _, lo, hi = self.buckets[i]
        self._step_range(lo, hi, self._slot_off[i], i)

    def _step_range(self, lo, hi, slot0, what):
        m = self._model()
        if m is not None and getattr(m, "backward_pending", None) is not None and m.backward_pending():
            # a split backward is in flight.  Stepping the [FCN + heads] bucket between its two phases is the documented use of
            # adam_step_bucket (backward_split(loss, between=...): those gradients are final after phase 1); anything that
            # touches the PointNet range, or any step before phase 1 ran at all, would use the previous step's gradients.
            cut = min(b[1] for b in self.buckets if b[0] != "pointnet") if len(self.buckets) > 1 else 0
            pend = getattr(m, "_pending_split", None)                # take_split() handed phase 2 over; phase 1 has run once
            phase1_done = (getattr(m, "_split", None) is None and pend is not None and pend.leaves is not None and
                           all(l.grad is not None for l in pend.leaves))      # the cut leaves hold their gradients
            if lo < cut or not phase1_done or len(self.buckets) == 1:
                raise RuntimeError("FlatTrainState: the model holds a half-finished split backward (forward with split_backward "
                                   "= True, phase 2 not differentiated yet): the PointNet gradients are the previous step's")
        if self.device.type != "cuda":
            raise RuntimeError("frustum_convnet_amd: the optimiser step is a HIP kernel (MI355X only); "
                               "there is no CPU fallback")
        L = _native.lib()
        off = 4 * lo                                     # bytes; bucket boundaries are 16-byte aligned
        if self.optimizer == "sgd":
            with torch.cuda.device(self.device):
                _native.check(L.fcn_sgd_step_f32(self.flat.data_ptr() + off, self.grad.data_ptr() + off,
                                                 self.exp_avg.data_ptr() + off, ctypes.c_int64(hi - lo),
                                                 self.hyper.data_ptr(), _native.current_stream(self.device)),
                              "fcn_sgd_step_f32")
            if lo == 0:
                self._step_slots[0:1] += 1          # (bookkeeping only: SGD has no bias correction)
            return
        with torch.cuda.device(self.device):
            _native.check(L.fcn_adam_step_f32(self.flat.data_ptr() + off, self.grad.data_ptr() + off,
                                              self.exp_avg.data_ptr() + off, self.exp_avg_sq.data_ptr() + off,
                                              ctypes.c_int64(hi - lo), self.hyper.data_ptr(),
                                              self._step_slots.data_ptr() + 8 * slot0,
                                              _native.current_stream(self.device)),
                          "fcn_adam_step_f32")

    def adam_step_scale(self, k):
        """The optimiser step of PointNet scale k (0-based) alone, on the current stream: the slice feat_net.pointnet<k+1>.* of the
        [pointnet] bucket.  A step loop that runs it right behind that scale's backward -- on the stream the scale's backward ran on --
        lets the scale's NEXT forward follow without waiting for the other scales (round 6 built that step: bit-identical, 2.5 % slower
        on MI355X, EXPERIMENTS 6.8 -- not in bench.py).  Every scale must then be stepped exactly once per training step and the
        [pointnet] bucket not at all; world 1 only (the gradients are not reduced)."""
        if self.comm:
            raise RuntimeError("FlatTrainState.adam_step_scale: per-scale steps are for a world of one rank (no all-reduce in between)")
        lo, hi = self.scale_ranges[k]
        assert lo % STEP_SPAN == 0
        self._step_range(lo, hi, lo // STEP_SPAN, "scale %d" % k)

    def adam_step(self):
        """The optimiser step of every bucket on the current stream: ONE launch over the whole buffer (the buckets exist for
        the gradient exchange; stepping them separately -- adam_step_bucket -- advances the same counters)."""
        self._step_range(0, self.numel, 0, "all")

    def step(self, zero_grad=False):
        """all-reduce + Adam.  The HIP backward kernels OVERWRITE every gradient they own, so no zero_grad is needed
        between steps when both fused paths are active; a model that routes some gradients through ordinary autograd
        (PointNetDet.fused_fcn = False, torch loss tail) ACCUMULATES into the same views -- pass zero_grad=True (or call
        zero_grad() before each forward) there.  Gradient accumulation over micro-batches is not supported by the fused
        backward (it overwrites)."""
        self.allreduce()
        self.adam_step()
        if zero_grad:
            self.zero_grad()

    # ---- checkpointing (reference: optimizer.state_dict() saved / restored at train/train_net_det.py:353,387)
    def state_dict(self):
        """torch.optim.Adam's state layout: {'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]} with i the
        position of the parameter in model.parameters() -- the order Adam(model.parameters()) of the reference indexes its
        state by, NOT the order of the flat buffer (which keeps the head tensors adjacent at its tail).  'names' lists the
        parameter names in that same order (an extra key torch ignores)."""
        step = int(self.step_count.item())
        hy = [float(v) for v in self.hyper.tolist()]
        state, names = {}, [None] * len(self.params)
        if self.optimizer == "sgd":           # torch.optim.SGD's layout: {'momentum_buffer': tensor} per parameter
            for k, (p, o) in enumerate(zip(self.params, self.offsets)):
                i = self.param_index[k]
                names[i] = self.names[k]
                state[i] = {"momentum_buffer": self.exp_avg[o:o + p.numel()].view(p.shape).clone()}
            group = {"lr": hy[0], "momentum": hy[1], "dampening": 0, "weight_decay": hy[2], "nesterov": False,
                     "params": list(range(len(self.params)))}
            return {"state": {i: state[i] for i in sorted(state)}, "param_groups": [group], "names": names, "step": step}
        for k, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            i = self.param_index[k]
            names[i] = self.names[k]
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        state = {i: state[i] for i in sorted(state)}
        group = {"lr": hy[0], "betas": (hy[1], hy[2]), "eps": hy[3], "weight_decay": hy[4], "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group], "names": names}

    def load_state_dict(self, sd):
        """Accepts this object's state_dict() and torch.optim.Adam(model.parameters()).state_dict() of the same model (the
        reference's checkpoints).  When 'names' is present the entries are matched BY NAME; every moment's shape is checked
        against its parameter before anything is copied."""
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("optimizer state has %d parameters, model has %d" % (len(group["params"]), len(self.params)))
        names = sd.get("names")
        if names is not None:
            if sorted(names) != sorted(self.names):
                raise ValueError("optimizer state names do not match the model's parameters")
            where = {n: i for i, n in enumerate(names)}
            index = [where[n] for n in self.names]
        else:
            index = self.param_index
        ids = group["params"]
        entries = []
        fields = ("momentum_buffer",) if self.optimizer == "sgd" else ("exp_avg", "exp_avg_sq")
        for k, p in enumerate(self.params):
            key = ids[index[k]]
            st = sd["state"].get(key, sd["state"].get(str(key)))
            if st is not None:
                for f in fields:
                    if st.get(f) is None:
                        # torch.optim.SGD keeps momentum_buffer = None until the first step (and with momentum 0): a zero
                        # buffer below; an Adam entry without its moments is not a state this optimiser can resume from
                        if self.optimizer == "sgd":
                            continue
                        raise ValueError("optimizer state entry %s (%s) has no '%s'" % (key, self.names[k], f))
                    if tuple(st[f].shape) != tuple(p.shape):
                        raise ValueError("optimizer state entry %s (%s): %s has shape %s, the parameter %s" % (
                            key, self.names[k], f, tuple(st[f].shape), tuple(p.shape)))
                if self.optimizer != "sgd" and "step" not in st:
                    raise ValueError("optimizer state entry of %s has no 'step'" % self.names[k])
            entries.append(st)
        # (everything is validated here, before the first copy: a rejected state_dict leaves this optimiser untouched)
        steps = set(int(float(st["step"])) for st in entries if st is not None and "step" in st) if self.optimizer != "sgd" else set()
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): the flat optimiser keeps one" % sorted(steps))
        if self.optimizer == "sgd":
            with torch.no_grad():
                for st, p, o in zip(entries, self.params, self.offsets):
                    n = p.numel()
                    if st is None or st.get("momentum_buffer") is None:
                        self.exp_avg[o:o + n].zero_()
                    else:
                        self.exp_avg[o:o + n].copy_(st["momentum_buffer"].reshape(-1))
                self._step_slots.fill_(int(sd.get("step", 0)))
                self.hyper[0:3].copy_(torch.tensor([group["lr"], group["momentum"], group["weight_decay"]],
                                                   dtype=torch.float32))
            return
        with torch.no_grad():
            for st, p, o in zip(entries, self.params, self.offsets):
                n = p.numel()
                if st is None:
                    self.exp_avg[o:o + n].zero_()
                    self.exp_avg_sq[o:o + n].zero_()
                    continue
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self._step_slots.fill_(steps.pop() if steps else 0)          # every workgroup's slot
            b1, b2 = group["betas"]
            self.hyper[0:5].copy_(torch.tensor([group["lr"], b1, b2, group["eps"], group["weight_decay"]],
                                               dtype=torch.float32))

    def release(self):
        """Back to ordinary autograd gradients (per-tensor .grad, accumulation)."""
        for p in self.params:
            if hasattr(p, "_fcn_grad"):
                del p._fcn_grad
            p.grad = None
