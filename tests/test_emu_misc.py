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
