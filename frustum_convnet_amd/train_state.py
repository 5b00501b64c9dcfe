"""Flat training state of the step loop (reference: train/train_net_det.py:114-137 -- forward, loss.mean(),
backward, optimizer.step() -- with optim.Adam from :321-339 and nn.DataParallel from :308-309).

Parameters, gradients and both Adam moments are four contiguous fp32 buffers (13.3 MB each for PointNetDet):
  * every nn.Parameter is re-homed as a view of `flat`, its .grad is a permanent view of `grad`;
  * the HIP backward kernels write each weight gradient STRAIGHT into its view (no autograd accumulation kernels,
    no per-tensor allocations) -- gradients are therefore overwritten, not accumulated, by every backward;
  * cls_out / reg_out weights (and biases) are adjacent, so the fused heads GEMM reads them as one matrix without a cat;
  * the data-parallel exchange is ONE RCCL all-reduce (sum) of `grad`; the 1/world mean is folded into the
    optimiser kernel's grad_scale;
  * the optimiser step is one streaming HIP kernel (fcn_adam_step_f32) whose step counter and hyper-parameters
    live in device memory, so it can be captured into the step's hipGraph and the learning rate changed between
    replays.
"""
import ctypes

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


class FlatTrainState:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world=1, group=None):
        named = _ordered_parameters(model)
        assert named, "model has no parameters"
        params = [p for _, p in named]
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatTrainState: fp32 parameters only")
        # every tensor starts on a 16-byte boundary (float4 access in the kernels, heads adjacency is preserved
        # because the head tensors' sizes are handled as one block below)
        offs, off = [], 0
        for n, p in named:
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
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, 1.0 / self.world], device=dev,
                                  dtype=torch.float32)
        # one step counter per workgroup of the optimiser kernel (all equal); step_count is slot 0
        nslot = int(_native.lib().fcn_adam_step_slots(ctypes.c_int64(total)))
        self._step_slots = torch.zeros(max(nslot, 1), device=dev, dtype=torch.int64)
        self.step_count = self._step_slots[0:1]
        self.device = dev

    def set_lr(self, lr):
        """Device-side update: takes effect on the next (eager or replayed) step."""
        self.hyper[0:1].fill_(float(lr))

    def zero_grad(self):
        self.grad.zero_()

    def allreduce(self):
        """Sum of the flat gradient over ranks (the mean's 1/world lives in hyper[5]).  world 1: no-op."""
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)

    def adam_step(self):
        if self.device.type != "cuda":
            raise RuntimeError("frustum_convnet_amd: the optimiser step is a HIP kernel (MI355X only); "
                               "there is no CPU fallback")
        L = _native.lib()
        with torch.cuda.device(self.device):
            _native.check(L.fcn_adam_step_f32(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), ctypes.c_int64(self.numel),
                                              self.hyper.data_ptr(), self._step_slots.data_ptr(),
                                              _native.current_stream(self.device)),
                          "fcn_adam_step_f32")

    def step(self):
        self.allreduce()
        self.adam_step()

    def release(self):
        """Back to ordinary autograd gradients (per-tensor .grad, accumulation)."""
        for p in self.params:
            if hasattr(p, "_fcn_grad"):
                del p._fcn_grad
            p.grad = None
