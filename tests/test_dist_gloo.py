"""N > 1 path on CPU: world_size-2 gloo processes exercise what bench.py's multi-GPU step uses -- the flat training state of
the REAL PointNetDet (parameter re-homing, the two gradient buckets cut at the FCN / PointNet boundary, the bucketed
asynchronous all-reduce and the single-call one, 1/world folded into the optimiser's grad_scale), the initial state broadcast
and the batch sharding.  The HIP kernels themselves need a GPU; the collective logic does not."""
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
    from frustum_convnet_amd import dist as fdist, det_base
    from frustum_convnet_amd.config import reset_cfg
    from frustum_convnet_amd.train_state import FlatTrainState
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    reset_cfg()
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    model = det_base.PointNetDet(3, num_vec=3, num_classes=2)
    fdist.broadcast_state(model, 0)
    st = FlatTrainState(model, lr=1e-3, world=world)
    w0 = st.flat.clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    q.put((rank, "same_init", all(torch.equal(gathered[0], g) for g in gathered)))
    # layout: every parameter / gradient is a view of the flat buffers; the buckets tile the buffer and are cut at the
    # FCN / PointNet boundary
    ok = all(p.data_ptr() == st.flat.data_ptr() + 4 * o and p.grad.data_ptr() == st.grad.data_ptr() + 4 * o
             for p, o in zip(st.params, st.offsets))
    names = [n for n, _, _ in st.buckets]
    spans = sorted((lo, hi) for _, lo, hi in st.buckets)
    ok = ok and names == ["fcn+heads", "pointnet"] and spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == st.numel
    cut = spans[0][1]
    for n, o, p in zip(st.names, st.offsets, st.params):
        inside_pn = o + p.numel() <= cut
        ok = ok and (inside_pn == n.startswith("feat_net."))
    q.put((rank, "layout", bool(ok)))
    # bucketed asynchronous all-reduce == mean of the ranks' local gradients (after the optimiser's 1/world)
    torch.manual_seed(11 + rank)
    st.grad.copy_(torch.randn(st.numel))
    local = st.grad.clone()
    st.allreduce_bucket_async(0)
    st.allreduce_bucket_async(1)
    st.wait_allreduce()
    allg = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allg, local)
    mean = sum(allg) / world
    q.put((rank, "bucketed", bool(torch.allclose(st.grad * float(st.hyper[5]), mean, rtol=1e-6, atol=1e-7))))
    # three pieces -- [FCN + heads], the wide PointNet scales, the narrow ones (bench.py's overlapped step) -- give the same: the
    # scale ranges tile the [pointnet] bucket
    sr = st.scale_ranges
    tiles = sorted(sr.values())
    q.put((rank, "scale_ranges", bool(sorted(sr) == [0, 1, 2, 3] and tiles[0][0] == 0 and tiles[-1][1] <= cut and
                                      all(a[1] <= b[0] for a, b in zip(tiles, tiles[1:])) and
                                      all(sr[k][0] <= o < sr[k][1] for n, o in zip(st.names, st.offsets)
                                          for k in range(4) if n.startswith("feat_net.pointnet%d." % (k + 1))))))
    st.grad.copy_(local)
    st.allreduce_bucket_async(0)
    st.allreduce_scales_async([2, 3])
    st.allreduce_scales_async([0, 1])
    st.wait_allreduce()
    used = torch.zeros(st.numel, dtype=torch.bool)
    for p_, o_ in zip(st.params, st.offsets):
        used[o_:o_ + p_.numel()] = True                # (alignment pad slots between the pieces belong to no parameter)
    q.put((rank, "pieces", bool(torch.allclose((st.grad * float(st.hyper[5]))[used], mean[used], rtol=1e-6, atol=1e-7))))
    # single-call path gives the same
    st.grad.copy_(local)
    st.allreduce()
    q.put((rank, "single", bool(torch.allclose(st.grad * float(st.hyper[5]), mean, rtol=1e-6, atol=1e-7))))
    # optimiser state round trip (resume): torch.optim.Adam-shaped dict
    st.exp_avg.copy_(torch.randn(st.numel)); st.exp_avg_sq.copy_(torch.rand(st.numel)); st._step_slots.fill_(7)
    sd = st.state_dict()
    st2 = FlatTrainState(det_base.PointNetDet(3, num_vec=3, num_classes=2), lr=5e-4, world=world)
    st2.load_state_dict(sd)
    sd2 = st2.state_dict()           # (the flat buffers also hold a few alignment pad slots that belong to no parameter)
    ok = all(torch.equal(sd2["state"][i]["exp_avg"], sd["state"][i]["exp_avg"]) and
             torch.equal(sd2["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"]) for i in range(len(st.params)))
    ok = ok and int(st2._step_slots.min()) == 7 and int(st2._step_slots.max()) == 7 and abs(float(st2.hyper[0]) - 1e-3) < 1e-9
    ok = ok and len(sd["state"]) == len(st.params) and sd["state"][0]["exp_avg"].shape == st.params[0].shape
    q.put((rank, "optim_state", bool(ok)))
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
        p.join(180)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(world * 7)]
    bad = [g for g in got if not g[2]]
    assert len(got) == 14 and {g[1] for g in got} >= {"scale_ranges", "pieces", "bucketed", "single"} and not bad, bad


def test_shard_batch_matches_dataparallel_scatter():
    from frustum_convnet_amd import dist as fdist
    b = {"a": torch.arange(24).view(8, 3), "b": torch.arange(8)}
    parts = [fdist.shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["a"] for p in parts]), b["a"])
    assert torch.equal(parts[2]["b"], torch.tensor([4, 5]))
    with pytest.raises(AssertionError):
        fdist.shard_batch(b, 0, 3)
