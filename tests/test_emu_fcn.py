"""CPU check of the FCN kernels THEMSELVES (not of the oracle): csrc/fcn_net.hip is compiled unmodified for the host against
the HIP stand-in of tests/host_harness/hip_emu (threads as coroutines, MFMA from the deposited operand registers) and its
forward + backward are compared with the nn.Conv1d / BatchNorm1d module path in fp64.  This pins index arithmetic, LDS
choreography, split-K reductions and role dispatch before any GPU minute is spent; the GPU parity tests stay the gate for
the real thing (hardware MFMA rounding, memory model, occupancy).  reference: models/det_base.py:163-224,250-258."""
import os
import shutil

import pytest

CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which(CLANG)), reason="host clang++ not available")


@pytest.fixture(scope="module")
def emu():
    from emu_fcn import load_emu
    return load_emu()


def _check(res, logit_tol, grad_tol):
    assert res["logits_abs"] < logit_tol and res["logits_pad"] == 0.0, res
    for k in ("dfeats", "dW", "dWh", "dgamma", "dbeta", "dbias"):
        assert res[k] < grad_tol, (k, res[k])
    assert res["rmean"] < 1e-5 and res["rvar"] < 1e-5, res


@pytest.mark.parametrize("precision,logit_tol,grad_tol", [(0, 1e-4, 1e-3), (1, 1e-5, 1e-4)])
def test_emulated_fcn_matches_module_path_4_levels(emu, precision, logit_tol, grad_tol):
    from emu_fcn import run_case
    res = run_case(emu, 2, [20, 10, 5, 3], nlev=4, precision=precision, verbose=False)
    _check(res, logit_tol, grad_tol)


def test_emulated_fcn_matches_module_path_5_levels_cropped(emu):
    """SUN-RGBD pyramid with odd lengths: every deconvolution output is cropped (38 / 40 / 40 -> 37 positions)."""
    from emu_fcn import run_case
    res = run_case(emu, 2, [37, 19, 10, 5, 3], nlev=5, precision=0, verbose=False)
    _check(res, 1e-4, 1e-3)


@pytest.mark.parametrize("precision,logit_tol,grad_tol", [(0, 1e-4, 1e-3), (1, 1e-5, 1e-4), (3, 6e-2, 1.5)])
def test_emulated_fcn_direct_forward_variant(precision, logit_tol, grad_tol):
    """-DFCN_FWD_DIRECT=1 (in the tree, off by default: EXPERIMENTS 6.2): the forward K-groups load their MFMA operands straight into
    fragments, no LDS staging.  Same comparison with the fp64 module path as the product kernels, in the split, the exact-fp32 (whose
    fragments go element by element into v_mfma_f32_32x32x2f32: the __builtin_bit_cast-on-a-vector-element trap) and the bf16 operand
    mode (loose bars: one bf16 term per product)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness"))
    import build_emu
    import emu_fcn
    path = build_emu.build(extra=["-DFCN_FWD_DIRECT=1"], out=os.path.join(build_emu.BUILD, "libfcn_emu_direct.so"))
    saved = emu_fcn.emu_path
    emu_fcn.emu_path = lambda force=False: path
    try:
        L = emu_fcn.load_emu()
    finally:
        emu_fcn.emu_path = saved
    res = emu_fcn.run_case(L, 2, [20, 10, 5, 3], nlev=4, precision=precision, verbose=False)
    assert res["logits_abs"] < logit_tol and res["logits_pad"] == 0.0, {k: v for k, v in res.items() if not k.startswith("_")}
    for k in ("dfeats", "dW", "dWh", "dgamma", "dbeta", "dbias"):
        assert res[k] < grad_tol, (k, res[k])
