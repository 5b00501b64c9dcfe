"""N > 1 path on CPU: world_size-2 gloo processes exercise the flat-parameter re-homing, the gradient
mean all-reduce (chunked and single), the state broadcast and the batch sharding used by bench.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from frustum_convnet_amd import dist as fdist
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    model = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3), torch.nn.BatchNorm1d(8), torch.nn.Conv1d(8, 2, 1))
    flat = fdist.FlatParams(model)
    fdist.broadcast_state(model, 0)
    w0 = flat.flat.clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    same_init = all(torch.equal(gathered[0], g) for g in gathered)
    # params are views of the flat buffer; grads accumulate into the flat grad buffer
    x = torch.randn(6, 4, 16, generator=torch.Generator().manual_seed(7))
    xs = fdist.shard_batch({"x": x}, rank, world)["x"]
    flat.zero_grad()
    model(xs).square().mean().backward()
    views_ok = all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in model.parameters())
    local = flat.grad.clone()
    for nb in (1, 3):
        flat.grad.copy_(local)
        fdist.GradAllReducer(flat, world, nbucket=nb).allreduce()
        allg = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(allg, local)
        mean = sum(allg) / world
        ok = torch.allclose(flat.grad, mean, rtol=1e-6, atol=1e-7)
        q.put((rank, nb, bool(ok), bool(same_init), bool(views_ok)))
    # per-tensor gradients through the coalesced reducer (what bench.py uses)
    model2 = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3), torch.nn.Conv1d(8, 2, 1))
    torch.manual_seed(5)
    for p_ in model2.parameters():
        p_.grad = torch.randn_like(p_) * (rank + 1)
    loc = [p_.grad.clone() for p_ in model2.parameters()]
    fdist.CoalescedGradAllReducer(list(model2.parameters()), world).allreduce()
    okc = True
    for p_, l_ in zip(model2.parameters(), loc):
        allg = [torch.zeros_like(l_) for _ in range(world)]
        dist.all_gather(allg, l_)
        okc = okc and torch.allclose(p_.grad, sum(allg) / world, rtol=1e-6, atol=1e-7)
    q.put((rank, "coalesced", bool(okc), True, True))
    # the step loop's flat training state: one summing all-reduce of the flat gradient, 1/world folded into hyper[5]
    from frustum_convnet_amd.train_state import FlatTrainState
    model3 = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3), torch.nn.BatchNorm1d(8), torch.nn.Conv1d(8, 2, 1))
    st = FlatTrainState(model3, lr=1e-3, world=world)
    torch.manual_seed(11 + rank)
    st.grad.copy_(torch.randn(st.numel))
    loc3 = st.grad.clone()
    st.allreduce()
    allg = [torch.zeros_like(loc3) for _ in range(world)]
    dist.all_gather(allg, loc3)
    ok3 = torch.allclose(st.grad * float(st.hyper[5]), sum(allg) / world, rtol=1e-6, atol=1e-7)
    ok3 = ok3 and all(p.grad.data_ptr() == st.grad.data_ptr() + 4 * o for p, o in zip(st.params, st.offsets))
    q.put((rank, "flat_state", bool(ok3), True, True))
    # optimizer over the flat parameter moves every view
    fp = flat.as_parameter()
    before = model[0].weight.detach().clone()
    torch.optim.SGD([fp], lr=0.1).step()
    q.put((rank, "moved", bool(not torch.equal(before, model[0].weight)), True, True))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(world * 5)]
    assert len(got) == 10 and all(g[2] and g[3] and g[4] for g in got), got


def test_shard_batch_matches_dataparallel_scatter():
    from frustum_convnet_amd import dist as fdist
    b = {"a": torch.arange(24).view(8, 3), "b": torch.arange(8)}
    parts = [fdist.shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["a"] for p in parts]), b["a"])
    assert torch.equal(parts[2]["b"], torch.tensor([4, 5]))
    with pytest.raises(AssertionError):
        fdist.shard_batch(b, 0, 3)
