"""-m gpu, needs >= 2 GPUs (skipped on a one-GPU box): two ranks over RCCL run the overlapped data-parallel step of
bench.py on the REAL PointNetDet (two-phase backward, bucketed asynchronous all-reduce, flat Adam) and check that
(1) the all-reduced gradient equals the mean of the two ranks' local gradients and (2) the parameters stay identical on both
ranks after three steps (reference semantics: nn.DataParallel's gradient reduce, train/train_net_det.py:308-309,120-128)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from frustum_convnet_amd import dist as fdist, synth
    from frustum_convnet_amd.train_state import FlatTrainState
    from helpers import load_golden, golden_inputs
    from test_gpu_model import _model
    r, w, local = fdist.init_from_env(backend="nccl")
    torch.cuda.set_device(local)
    g = load_golden("car_b4_n512")
    full = golden_inputs(g)
    data = synth.to_torch({k: v[rank * 2:(rank + 1) * 2] for k, v in full.items()}, "cuda")     # 2 frustums per rank
    m = _model(g)
    m.train()
    m.split_backward = True
    fdist.broadcast_state(m, 0)
    st = FlatTrainState(m, lr=1e-4, weight_decay=1e-4, world=world)
    ok_grad = True
    for it in range(3):
        lo, _ = m(data)
        m.backward_split(lo["total_loss"], between=lambda: st.allreduce_bucket_async(0))
        local_grad_pn = st.grad[st.buckets[1][1]:st.buckets[1][2]].clone()         # PointNet bucket: not yet reduced
        st.allreduce_bucket_async(1)
        st.wait_allreduce()
        torch.cuda.synchronize()
        both = [torch.zeros_like(local_grad_pn) for _ in range(world)]
        dist.all_gather(both, local_grad_pn)
        mean = sum(both) / world
        got = st.grad[st.buckets[1][1]:st.buckets[1][2]] * float(st.hyper[5])
        ok_grad = ok_grad and bool(torch.allclose(got, mean, rtol=1e-5, atol=1e-8))
        st.adam_step()
    torch.cuda.synchronize()
    flats = [torch.zeros_like(st.flat) for _ in range(world)]
    dist.all_gather(flats, st.flat)
    q.put((rank, ok_grad, bool(torch.equal(flats[0], flats[1]))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_two_rank_overlapped_step_over_rccl():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(world)]
    assert all(g[1] and g[2] for g in got), got
