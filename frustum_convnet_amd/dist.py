"""One-process-per-GPU data parallelism for the hot path (replaces nn.DataParallel,
train/train_net_det.py:308-309).

Frustums are independent except for the gradient sum, so each rank keeps B_local frustums, per-rank
BatchNorm statistics (exactly what DataParallel replicas do -- the reference has no SyncBN) and the only
exchange step is the all-reduce (mean) of the 3.3 M-parameter gradient over RCCL/xGMI.  Parameters and
gradients live in flat fp32 buffers (train_state.FlatTrainState), so the exchange is two large collectives cut where
the backward finishes them -- [ConvFeatNet + heads] 11.6 MB while the PointNet backward still runs, then [PointNet]
1.1 MB -- instead of 154 small ones (fully-connected xGMI: RCCL drives all 7 links with one large message).
This module holds the process-group plumbing only: rank discovery, initial broadcast, batch sharding.
"""
import os

import torch
import torch.distributed as dist


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_from_env(backend=None, single_rank_group=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local).
    A world of one creates no process group -- unless single_rank_group: then a ONE-rank group of `backend` is formed (rendezvous
    on 127.0.0.1 when the launcher set none): a 1-rank "nccl" group is a real RCCL communicator on a one-GPU box, which is how
    the N > 1 step (communication stream, captured collectives) is exercised without a second GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        if world == 1 and "MASTER_ADDR" not in os.environ:
            dist.init_process_group(backend=backend, rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % _free_port())
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


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
