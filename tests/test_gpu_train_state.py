"""-m gpu: the step loop's flat training state (train_state.FlatTrainState + C-ABI fcn_adam_step_f32) against
torch.optim.Adam on the same gradients, and the 2-step loss trajectory against the CPU oracle stepped with
torch.optim.Adam (reference loop: train/train_net_det.py:114-137, optimiser :321-339)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, golden_inputs, golden_state_dict
from frustum_convnet_amd import synth
from test_gpu_model import _model

pytestmark = pytest.mark.gpu


def test_flat_layout_and_direct_gradients():
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    ref = _model(g)
    ref.train()
    lo, _ = ref(data)
    lo["total_loss"].backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}

    m = _model(g)
    m.train()
    st = FlatTrainState(m, lr=1e-3, weight_decay=1e-4)
    named = dict(m.named_parameters())
    assert st.numel >= sum(p.numel() for p in named.values())
    for k, p in named.items():
        assert p.grad is p._fcn_grad and p.grad.data_ptr() % 4 == 0
        assert st.flat.data_ptr() <= p.data_ptr() < st.flat.data_ptr() + 4 * st.numel
    cw, rw = named["cls_out.weight"], named["reg_out.weight"]
    assert rw.data_ptr() == cw.data_ptr() + 4 * cw.numel()          # heads adjacent: one GEMM operand, no cat
    st.grad.fill_(float("nan"))                                      # every element must be overwritten
    lo2, _ = m(data)
    lo2["total_loss"].backward()
    # (the row-major loss tail sums its workgroup partials in a fixed order: the scalar is reproducible bit for bit)
    assert float(lo2["total_loss"]) == float(lo["total_loss"])
    for k, p in named.items():
        assert p.grad is p._fcn_grad                                 # autograd did not replace the view
        assert torch.equal(p.grad, want[k]), k                       # same kernels, same bits, written in place
    # a second backward overwrites (no accumulation)
    lo3, _ = m(data)
    lo3["total_loss"].backward()
    k = "conv_net.block1_conv1.0.weight"
    assert float((named[k].grad - want[k]).abs().max()) <= 1e-4 * float(want[k].abs().max())
    used = torch.zeros(st.numel, dtype=torch.bool, device="cuda")
    for p, o in zip(st.params, st.offsets):
        used[o:o + p.numel()] = True
    assert torch.isfinite(st.grad[used]).all()


def test_flat_adam_matches_torch_adam():
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    a = _model(g)
    b = _model(g)
    a.train()
    b.train()
    opt = torch.optim.Adam(a.parameters(), lr=1e-3, weight_decay=1e-4)
    st = FlatTrainState(b, lr=1e-3, weight_decay=1e-4)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        la, _ = a(data)
        la["total_loss"].backward()
        opt.step()
        lb, _ = b(data)
        lb["total_loss"].backward()
        st.step()
        # the third forward already sees parameters that went through one ill-conditioned update (see below): the
        # two fp32 trajectories land on 67.76 and 66.85 (the CPU oracle gives 67.76 in fp32 and 66.85 in fp64)
        rel = 2e-4 if it < 2 else 5e-2
        assert abs(float(la["total_loss"]) - float(lb["total_loss"])) <= rel * abs(float(la["total_loss"])), it
        worst = max(float((pa[k] - pb[k]).abs().max()) for k in pa)
        print("flat Adam vs torch.optim.Adam after step %d: max |dparam| %.2e" % (it + 1, worst))
        for k in pa:
            d = float((pa[k] - pb[k]).abs().max())
            if it == 0:
                # identical gradients in, one update: only the fp32 rounding of the update arithmetic differs
                assert d <= 1e-7 + 1e-6 * float(pa[k].abs().max()), (k, d)
            else:
                # later steps see slightly different gradients (lr 1e-3 on these weights is an unstable regime: the
                # loss goes 106 -> 249 -> 67) and Adam's m / sqrt(v) is ill-conditioned where a gradient element
                # changes sign, so single elements may differ by up to the step size (~lr per step) while the
                # tensors as a whole stay together
                assert d <= 3.5e-3, (k, d, it)
                assert float((pa[k] - pb[k]).abs().mean()) <= 3e-4, (k, it)
    assert int(st.step_count) == 3
    # learning-rate change on the device
    st.set_lr(0.0)
    before = st.flat.clone()
    st.adam_step()
    assert torch.equal(before, st.flat) and int(st.step_count) == 4


def test_buckets_stepped_separately_and_sgd_matches_torch():
    """ adam_step_bucket(i) alone, in either order, equals adam_step(); and the 'sgd' optimiser of the reference's
    step loop (train/train_net_det.py:325-327) against torch.optim.SGD on the same gradients."""
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    models = [_model(g) for _ in range(3)]
    states = [FlatTrainState(m, lr=1e-4, weight_decay=1e-4) for m in models[:2]]
    for m in models:
        m.train()
    for it in range(2):
        for m in models[:2]:
            lo, _ = m(data)
            m.backward(lo["total_loss"])
        states[0].adam_step()
        states[1].adam_step_bucket(1)                 # PointNet bucket first, then [FCN + heads]
        states[1].adam_step_bucket(0)
    torch.cuda.synchronize()
    assert torch.equal(states[0].flat, states[1].flat) and int(states[1]._step_slots.min()) == 2
    # SGD + momentum
    a, b = models[2], _model(g)
    b.train()
    opt = torch.optim.SGD(a.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    st = FlatTrainState(b, lr=1e-4, weight_decay=1e-4, optimizer="sgd", momentum=0.9)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        la, _ = a(data)
        la["total_loss"].backward()
        opt.step()
        lb, _ = b(data)
        b.backward(lb["total_loss"])
        st.step()
    worst = max(float((pa[k] - pb[k]).abs().max()) for k in pa)
    print("flat SGD vs torch.optim.SGD after 2 steps: max |dparam| %.2e" % worst)
    assert worst < 2e-5
    sd = st.state_dict()
    opt2 = torch.optim.SGD(a.parameters(), lr=1e-3, momentum=0.5)
    opt2.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})      # torch's layout, model order
    # (the second step's gradients already differ a little between the two trajectories: the bound is relative to the tensor)
    wb = max(float((opt2.state[p]["momentum_buffer"] - opt.state[p]["momentum_buffer"]).abs().max() /
                   (opt.state[p]["momentum_buffer"].abs().max() + 1e-12)) for p in a.parameters())
    print("flat SGD momentum buffers vs torch after 2 steps: worst max-relative difference %.2e" % wb)
    assert wb < 2e-2


@pytest.mark.parametrize("lr", [1e-3, 1e-4])
def test_two_step_trajectory_vs_cpu_oracle(lr):
    """Loss after 0, 1 and 2 Adam(lr, wd 1e-4) steps from the same weights: HIP step loop vs oracle/det_ref.py stepped by
    torch.optim.Adam on the CPU.  The referee is the oracle evaluated in fp64.
    lr 1e-3 (the reference's BASE_LR) is an unstable regime with these synthetic weights (106.51 -> 248.93 -> 66.85): the
    first two values are held to rel 1e-3, but the third amplifies fp32 rounding -- the oracle's own fp32 evaluation lands
    on 67.76 (1.4 % off its fp64 value, measured) and so does this path on some builds -- so it only gets a 5 % bound.
    lr 1e-4 (stable) holds all three values to rel 1e-3."""
    from oracle import det_ref
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data_np = golden_inputs(g)
    strides = tuple(g["meta_strides"])
    f64 = lambda v: v.double() if v.dtype.is_floating_point else v
    sd = {k: f64(v.clone()) for k, v in golden_state_dict(g).items()}
    leaves = []
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
            leaves.append(v)
    opt = torch.optim.Adam(leaves, lr=lr, weight_decay=1e-4)
    dcpu = {k: f64(v) for k, v in synth.to_torch(data_np).items()}
    ref_losses = []
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        _, _, lo = det_ref.forward(sd, dcpu, strides, training=True)    # train mode: batch statistics only
        ref_losses.append(float(lo["total_loss"]))
        lo["total_loss"].backward()
        opt.step()
    m = _model(g)
    m.train()
    st = FlatTrainState(m, lr=lr, weight_decay=1e-4)
    data = synth.to_torch(data_np, "cuda")
    got = []
    for it in range(3):
        lo, _ = m(data)
        got.append(float(lo["total_loss"]))
        lo["total_loss"].backward()
        st.step()
    print("trajectory lr", lr, "fp64 oracle", ref_losses, "hip", got)
    for it, (r, h) in enumerate(zip(ref_losses, got)):
        tol = 5e-2 if (lr > 5e-4 and it == 2) else 1e-3
        assert abs(r - h) <= tol * abs(r), (lr, ref_losses, got)


def test_short_training_run_in_one_hipgraph():
    """Ten optimiser steps with the whole step (forward, backward, Adam) captured once and replayed -- what bench.py
    times: the loss falls, the step counter advances on the device, and the replayed graph gives the same parameters as
    the same ten steps launched eagerly."""
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")

    def make():
        m = _model(g)
        m.train()
        return m, FlatTrainState(m, lr=1e-4, weight_decay=1e-4)

    m1, s1 = make()
    eager = []
    for _ in range(10):
        lo, _ = m1(data)
        eager.append(float(lo["total_loss"]))
        lo["total_loss"].backward()
        s1.step()
    assert eager[-1] < 0.6 * eager[0] and int(s1.step_count) == 10

    m2, s2 = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture (allocator, workspaces)
        lo, _ = m2(data)
        lo["total_loss"].backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lo, _ = m2(data)
        lo["total_loss"].backward()
        s2.adam_step()
    # the warm-up forward moved the BN running statistics once more than the eager run; parameters are what we compare
    losses = []
    for _ in range(10):
        graph.replay()
        losses.append(float(lo["total_loss"]))
    assert int(s2.step_count) == 10
    assert abs(losses[-1] - eager[-1]) <= 2e-3 * abs(eager[-1]), (losses, eager)
    d = float((s1.flat - s2.flat).abs().max())
    assert d <= 5e-4, d


def test_two_graph_overlapped_step_equals_single_graph():
    """The N > 1 step of bench.py -- graph A (forward, loss, FCN backward) | bucket all-reduce | graph B (PointNet backward) |
    bucket all-reduce | Adam -- replayed at world size 1 (the collectives are no-ops) gives bit-identical parameters to the
    single-graph step after several steps: the cut at the pooled feature maps changes the launch structure, not the math."""
    from frustum_convnet_amd.train_state import FlatTrainState
    from frustum_convnet_amd.loss_fused import unit_grad
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")

    def make(split):
        m = _model(g)
        m.train()
        m.split_backward = split
        return m, FlatTrainState(m, lr=1e-4, weight_decay=1e-4)

    def warm(m):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            lo, _ = m(data)
            m.backward(lo["total_loss"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    m1, s1 = make(False)
    warm(m1)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        lo, _ = m1(data)
        m1.backward(lo["total_loss"])
        s1.adam_step()
    m2, s2 = make(True)
    warm(m2)
    # (round 4: the PointNet backward itself is cut in two -- wide scales, then narrow scales -- so that the wide scales' gradients
    # can be exchanged while the narrow ones are computed: three graphs)
    gA, gB, gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA):
        lo2, _ = m2(data)
        pending = m2.take_split()
        lo2["total_loss"].backward(gradient=unit_grad(lo2["total_loss"].device))
        assert m2.backward_pending()
    with torch.cuda.graph(gB, pool=gA.pool()):
        pending.backward(scales=[2, 3])
        assert m2.backward_pending()
        with pytest.raises(RuntimeError):
            pending.backward(scales=[3])              # a scale is differentiated once
    with torch.cuda.graph(gC, pool=gA.pool()):
        pending.backward(scales=[0, 1])
    assert not m2.backward_pending()
    assert [n for n, _, _ in s2.buckets] == ["fcn+heads", "pointnet"] and sorted(s2.scale_ranges) == [0, 1, 2, 3]
    for _ in range(5):
        g1.replay()
        gA.replay()
        s2.allreduce_bucket_async(0)
        gB.replay()
        s2.allreduce_scales_async([2, 3])
        gC.replay()
        s2.allreduce_scales_async([0, 1])
        s2.wait_allreduce()
        s2.adam_step()
    torch.cuda.synchronize()
    assert int(s1.step_count) == 5 and int(s2.step_count) == 5
    assert torch.equal(s1.grad, s2.grad)
    assert torch.equal(s1.flat, s2.flat)


def test_early_bucket_step_between_the_two_backward_phases():
    """ The documented early step: adam_step_bucket(0) on the [FCN + heads] bucket inside backward_split's
    between() -- its gradients are final after phase 1 -- then the PointNet bucket after phase 2: same parameters as one
    adam_step() after the whole backward.  The PointNet bucket (or the whole buffer) between the phases is refused."""
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    ma, mb = _model(g), _model(g)
    for m in (ma, mb):
        m.train()
    mb.split_backward = True
    sa, sb = FlatTrainState(ma, lr=1e-4, weight_decay=1e-4), FlatTrainState(mb, lr=1e-4, weight_decay=1e-4)
    assert [n for n, _, _ in sb.buckets] == ["fcn+heads", "pointnet"]
    refused = []

    def between():
        for bad in (lambda: sb.adam_step_bucket(1), sb.adam_step):
            try:
                bad()
                refused.append(False)
            except RuntimeError:
                refused.append(True)
        sb.adam_step_bucket(0)

    for it in range(2):
        la, _ = ma(data)
        ma.backward(la["total_loss"])
        sa.adam_step()
        lb, _ = mb(data)
        mb.backward_split(lb["total_loss"], between=between)
        sb.adam_step_bucket(1)
    torch.cuda.synchronize()
    assert refused == [True] * 4
    assert torch.equal(sa.flat, sb.flat) and int(sb._step_slots.min()) == 2
    # before phase 1 has run at all, even the [FCN + heads] bucket is refused
    lb, _ = mb(data)
    pending = mb.take_split()
    with pytest.raises(RuntimeError):
        sb.adam_step_bucket(0)
    lb["total_loss"].backward()
    pending.backward()
    sb.adam_step()


def test_load_state_dict_accepts_torch_sgd_without_momentum_buffers():
    """ torch.optim.SGD keeps momentum_buffer = None until its first step: such a checkpoint loads as zero buffers;
    an Adam entry without its moments is refused with a ValueError (not a KeyError / AttributeError)."""
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    m = _model(g)
    st = FlatTrainState(m, lr=1e-4, weight_decay=1e-4, optimizer="sgd", momentum=0.9)
    st.exp_avg.fill_(3.0)
    n = len(st.params)
    sd = {"state": {i: {"momentum_buffer": None} for i in range(n)},
          "param_groups": [{"lr": 1e-3, "momentum": 0.9, "dampening": 0, "weight_decay": 1e-4, "nesterov": False,
                            "params": list(range(n))}]}
    st.load_state_dict(sd)
    assert all(float(st.exp_avg[o:o + p.numel()].abs().max()) == 0.0 for p, o in zip(st.params, st.offsets))
    m2 = _model(g)
    st2 = FlatTrainState(m2, lr=1e-4, weight_decay=1e-4)
    sd2 = st2.state_dict()
    del sd2["state"][0]["exp_avg"]
    with pytest.raises(ValueError):
        st2.load_state_dict(sd2)


def test_scales_stepped_one_by_one_equal_the_bucket_step():
    """FlatTrainState.adam_step_scale(k): every PointNet scale starts on a multiple of the optimiser kernel's workgroup span, so stepping
    the scales one by one (+ the [FCN + heads] bucket) uses the same workgroups and step-counter slots as the whole-buffer launch --
    parameters, moments and counters bit for bit.  (Round 6 built a step on it -- each scale stepped behind ITS backward, its next
    forward chained behind that inside the captured graph: bit-identical and 2.5 % slower, EXPERIMENTS 6.8.)"""
    from frustum_convnet_amd.train_state import FlatTrainState, STEP_SPAN
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    models = [_model(g) for _ in range(2)]
    states = [FlatTrainState(m, lr=1e-4, weight_decay=1e-4) for m in models]
    assert sorted(states[0].scale_ranges) == [0, 1, 2, 3]
    assert all(lo % STEP_SPAN == 0 and hi <= states[0].buckets[1][2] for lo, hi in states[0].scale_ranges.values())
    for m in models:
        m.train()
    for it in range(3):
        for m in models:
            lo, _ = m(data)
            m.backward(lo["total_loss"])
        states[0].adam_step()
        for k in (2, 0, 3, 1):
            states[1].adam_step_scale(k)
        states[1].adam_step_bucket(0)
    torch.cuda.synchronize()
    assert int(states[1]._step_slots.min()) == int(states[1]._step_slots.max()) == 3
    assert torch.equal(states[0].flat, states[1].flat) and torch.equal(states[0].exp_avg_sq, states[1].exp_avg_sq)


def test_shared_backward_stream_gives_bit_identical_steps():
    """PointNetFeat.share_backward_stream(scale, host): a narrow scale's backward enqueued on another scale's stream, behind that scale's
    chain (one parallel branch less for the graph executor's four streams).  Eager (event edges both ways) and inside a captured
    two-step graph (stream order only): gradients after one backward and parameters after several replayed steps equal the plain
    model's bit for bit; a host that is not ahead of the guest in autograd's order is refused."""
    from frustum_convnet_amd.train_state import FlatTrainState
    g = load_golden("car_b4_n512")
    data = synth.to_torch(golden_inputs(g), "cuda")
    grads, flats = [], []
    for share in (None, (0, 2), (1, 2)):
        m = _model(g)
        m.train()
        s = FlatTrainState(m, lr=1e-4, weight_decay=1e-4)
        assert m.feat_net.bwd_share == {1: 2}                     # the 4-scale default
        m.feat_net.bwd_share.clear()
        if share is not None:
            m.feat_net.share_backward_stream(share[0], share[1])
        lo, _ = m(data)
        m.backward(lo["total_loss"])
        torch.cuda.synchronize()
        grads.append(s.grad.clone())
        s.adam_step()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(2):
                lo, _ = m(data)
                m.backward(lo["total_loss"])
                s.adam_step()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        flats.append(s.flat.clone())
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert torch.equal(flats[0], flats[1]) and torch.equal(flats[0], flats[2])
    with pytest.raises(ValueError):
        m.feat_net.share_backward_stream(2, 1)          # the host's node would run AFTER the guest's
    with pytest.raises(ValueError):
        m.feat_net.share_backward_stream(0, 3)          # the widest scale runs on the caller's stream
