"""-m gpu: every stage of the HIP PointNet scale (through the C-ABI) against the CPU emulation of the
entry-space algorithm (tests/entry_ref.py, itself proven equal to the dense oracle on CPU) and against the
dense oracle's pooled features.  Tolerance: 2e-4 of each tensor's max magnitude (fp32 summation order)."""
import pytest
import torch

import gpu_stage_check as gsc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gsc.CASES, ids=lambda c: "B%d_N%d_s%s_K%d_C%d" % (c[0], c[1], c[2], c[3], c[4][2]))
def test_stages(case):
    res = gsc.run_stages(*case, verbose=True)
    bad = gsc.check(res)
    assert not bad, bad


def test_uniform_variant_full_windows():
    res = gsc.run_stages(2, 512, 2.0, 8, (64, 64, 128), 2.0, variant="uniform")
    assert not gsc.check(res)
