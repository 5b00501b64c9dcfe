"""CPU checks of small kernels through the C-ABI of the host-emulated library (tests/host_harness): the optimiser step
(csrc/optim.hip) against torch.optim.Adam.  reference: train/train_net_det.py:321-339 (optim.Adam(lr, weight_decay))."""
import ctypes
import os
import shutil

import numpy as np
import pytest
import torch

CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which(CLANG)), reason="host clang++ not available")


@pytest.mark.parametrize("n", [4, 1000, 70000])        # below one workgroup, ragged, several workgroups
def test_emulated_adam_matches_torch_adam(n):
    from emu_fcn import emu_path
    L = ctypes.CDLL(emu_path())
    L.fcn_adam_step_slots.restype = ctypes.c_int64
    L.fcn_adam_step_slots.argtypes = [ctypes.c_int64]
    L.fcn_adam_step_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_void_p] * 3
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    lr, b1, b2, eps, wd, gscale = 1e-3, 0.9, 0.999, 1e-8, 1e-4, 0.5
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    hyper = torch.tensor([lr, b1, b2, eps, wd, gscale], dtype=torch.float32)
    slots = torch.zeros(max(int(L.fcn_adam_step_slots(n)), 1), dtype=torch.int64)
    for step in range(3):
        grad = torch.randn(n, generator=g)
        ref.grad = (grad.double() * gscale)
        opt.step()
        rc = L.fcn_adam_step_f32(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(),
                                 slots.data_ptr(), None)
        assert rc == 0
        assert int(slots.min()) == step + 1 and int(slots.max()) == step + 1        # every workgroup advanced its counter
        err = float((p.double() - ref.detach()).abs().max())
        assert err < 2e-6, (step, err)


@pytest.mark.parametrize("n", [4, 1001, 70000])
def test_emulated_sgd_matches_torch_sgd(n):
    """fcn_sgd_step_f32 against torch.optim.SGD(momentum, weight_decay) -- the 'sgd' branch of train/train_net_det.py:325-327."""
    from emu_fcn import emu_path
    L = ctypes.CDLL(emu_path())
    L.fcn_sgd_step_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 2
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    lr, mu, wd, gscale = 1e-2, 0.9, 1e-4, 0.5
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.SGD([ref], lr=lr, momentum=mu, weight_decay=wd)
    p, buf = p0.clone(), torch.zeros(n)
    hyper = torch.tensor([lr, mu, wd, gscale], dtype=torch.float32)
    for step in range(4):
        grad = torch.randn(n, generator=g)
        ref.grad = grad.double() * gscale
        opt.step()
        assert L.fcn_sgd_step_f32(p.data_ptr(), grad.data_ptr(), buf.data_ptr(), n, hyper.data_ptr(), None) == 0
        assert float((p.double() - ref.detach()).abs().max()) < 2e-6, step
        assert float((buf.double() - opt.state[ref]["momentum_buffer"]).abs().max()) < 2e-6, step
    assert L.fcn_sgd_step_f32(p.data_ptr(), grad.data_ptr(), buf.data_ptr(), 3, hyper.data_ptr(), None) != 0     # n < 4 refused


def test_lr_schedule_matches_torch_schedulers():
    """lr_for_epoch = StepLR / MultiStepLR as train/train_net_det.py:334-339 builds them + the MIN_LR clamp of :98-103."""
    from frustum_convnet_amd.train_state import lr_for_epoch
    for steps in ([20], [20, 35], [3, 5, 9]):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=1e-3)
        sch = (torch.optim.lr_scheduler.MultiStepLR(opt, milestones=steps, gamma=0.1) if len(steps) > 1 else
               torch.optim.lr_scheduler.StepLR(opt, step_size=steps[0], gamma=0.1))
        for epoch in range(50):
            want = max(opt.param_groups[0]["lr"], 1e-5)
            assert abs(lr_for_epoch(epoch, 1e-3, steps, 0.1, 1e-5) - want) < 1e-12, (steps, epoch)
            opt.step()
            sch.step()


def test_smoke_entry_under_emulation():
    """__graft_entry__.smoke() (the driver's first call on the GPU box) end to end on the CPU: same code path, emulated kernels."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as entry
    from emu_shim import emulated_gpu
    saved = torch.cuda.is_available
    with emulated_gpu():
        torch.cuda.is_available = lambda: True
        try:
            entry.smoke()
        finally:
            torch.cuda.is_available = saved


def test_prefetch_made_inside_a_capture_is_never_consumed_outside_it(monkeypatch):
    """a front prefetched while a hipGraph is being captured exists only inside that capture (its launches may never have
    run).  An eager forward on the same tensors, or another capture, must discard it -- not run phase 2 on workspaces nobody
    filled -- and must not wait for its event; the same capture consumes it.  Also: the key covers what the prepared handles
    froze (precision, parameter storage), so a change between prefetch() and forward() drops the prefetch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from emu_shim import emulated_gpu
    from frustum_convnet_amd import _native, synth, precision
    import bench
    with emulated_gpu():
        dev = torch.device("cuda", 0)
        torch.manual_seed(3)
        model = bench.build_model(dev, "car")
        model.train()
        data = synth.to_torch(synth.make_batch(2, 256, seed=11, variant="car", tilt=(0.01, 0.05)), dev)
        fn = model.feat_net
        cap = [0]
        monkeypatch.setattr(_native, "capture_id", lambda device=None: cap[0])

        def run():
            losses, _ = model(data)
            return float(losses["total_loss"])

        ref = run()                                        # plain step (the running statistics move; the loss of the SAME weights is compared below)
        sd = {k: v.clone() for k, v in model.state_dict().items()}

        def fresh():
            model.load_state_dict(sd)

        fresh(); base = run()
        # 1. prefetched inside capture 7, consumed by an eager forward: dropped, the full front runs -> same loss
        fresh(); cap[0] = 7
        assert model.prefetch(data) and fn._prefetched is not None and fn._prefetched["cap"] == 7
        cap[0] = 0
        assert fn._prefetch_is_foreign()
        waited = []
        ev = fn._prefetched["event"]
        assert run() == base and fn._prefetched is None
        # 2. ... by ANOTHER capture: dropped as well
        fresh(); cap[0] = 7; model.prefetch(data); cap[0] = 8
        assert fn._prefetch_is_foreign() and run() == base
        # 3. the same capture consumes it (phase 2 only): identical result
        fresh(); cap[0] = 9; model.prefetch(data)
        assert not fn._prefetch_is_foreign()
        assert run() == base and fn._prefetched is None
        cap[0] = 0
        # 3b. capturing, but the id cannot be determined: _native.capture_id hands out a FRESH negative number each time,
        # so a prefetch tagged with one never equals a later answer -- always foreign, dropped, never consumed
        unknown = [0]

        def unknown_id(device=None):
            unknown[0] -= 1
            return unknown[0]
        monkeypatch.setattr(_native, "capture_id", unknown_id)
        fresh(); model.prefetch(data)
        assert fn._prefetched["cap"] < 0 and fn._prefetch_is_foreign()
        assert run() == base and fn._prefetched is None
        monkeypatch.setattr(_native, "capture_id", lambda device=None: cap[0])
        # 4. a precision change between prefetch and forward: the handles froze the old code -> dropped, not consumed
        fresh(); model.prefetch(data)
        key = fn._prefetched["key"]
        with precision.precision("f32"):
            xyz = data["point_cloud"][:, :3, :].contiguous()
            refs = [data["center_ref%d" % i] for i in range(1, 5)]
            assert fn._front_key(model._pf_xyz[1], refs, data["one_hot"], True, True) != key
        fn.drop_prefetch()
        assert fn._prefetched is None
