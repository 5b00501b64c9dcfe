"""One-process-per-GPU data parallelism for the hot path (replaces nn.DataParallel,
train/train_net_det.py:308-309).

Frustums are independent except for the gradient sum, so each rank keeps B_local frustums, per-rank
BatchNorm statistics (exactly what DataParallel replicas do -- the reference has no SyncBN) and the only
exchange step is one all-reduce (mean) of the 3.3 M-parameter gradient over RCCL/xGMI.  Parameters and
gradients live in two flat fp32 buffers so that exchange is a single 13.3 MB collective (fully-connected
xGMI: RCCL can drive all 7 links at once with one large message, instead of 154 small ones).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatParams:
    """Re-homes every parameter (and its .grad) of `model` as a view into one contiguous buffer."""

    def __init__(self, model):
        params = [p for p in model.parameters()]
        assert params, "model has no parameters"
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.grad = torch.zeros(total, device=dev, dtype=dt)
        self.params = params
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += n
        self.numel = total

    def zero_grad(self):
        self.grad.zero_()

    def as_parameter(self):
        """A single leaf Parameter over the flat buffer whose .grad is the flat gradient (for the optimizer)."""
        fp = torch.nn.Parameter(self.flat, requires_grad=True)
        fp.grad = self.grad
        return fp


class GradAllReducer:
    """Gradient mean over ranks; world_size 1 is a no-op.  Optionally split into `nbucket` chunks launched
    back to back (each a separate RCCL call) so the tail of one overlaps the head of the next."""

    def __init__(self, flat, world, nbucket=1, group=None):
        self.flat, self.world, self.group = flat, world, group
        n = flat.numel
        edges = [n * i // nbucket for i in range(nbucket + 1)]
        self.chunks = [(edges[i], edges[i + 1]) for i in range(nbucket) if edges[i + 1] > edges[i]]

    def allreduce(self):
        if self.world == 1:
            return
        for a, b in self.chunks:
            dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
        self.flat.grad.div_(self.world)


class CoalescedGradAllReducer:
    """Gradient mean over ranks for per-tensor gradients: one flatten (single cat kernel), ONE RCCL all-reduce of the
    13.3 MB buffer, one scatter back.  world_size 1 is a no-op."""

    def __init__(self, params, world, group=None):
        self.params, self.world, self.group = list(params), world, group
        self.buf = None

    def allreduce(self):
        if self.world == 1:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        off = 0
        views = []
        for g in grads:
            n = g.numel()
            views.append(flat[off:off + n].view_as(g))
            off += n
        torch._foreach_copy_(grads, views)


def broadcast_state(model, src=0):
    """Rank `src`'s parameters and buffers to everyone (start of training, or DataParallel-like buffer sync)."""
    if not dist.is_initialized():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)


def shard_batch(batch, rank, world):
    """Contiguous split of a dict of (B, ...) tensors along dim 0: what DataParallel's scatter does."""
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        assert B % world == 0, "batch %d not divisible by world size %d" % (B, world)
        n = B // world
        out[k] = v[rank * n:(rank + 1) * n]
    return out
