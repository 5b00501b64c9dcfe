"""CPU check of the PointNet-scale kernels THEMSELVES: grouping, compaction, the fused conv GEMMs with their BatchNorm
statistics, pooling, and the whole backward (csrc/grouping.hip, pointnet_fwd.hip, pointnet_bwd.hip) compiled unmodified for
the host (tests/host_harness) and run through the C-ABI on CPU tensors -- the same stage-by-stage comparison as the GPU test
(tests/test_gpu_pointnet.py -> gpu_stage_check.run_stages) against the entry-space reference and the dense oracle.
reference: models/det_base.py:35-103,126-159; ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-65."""
import os
import shutil

import pytest

CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which(CLANG)), reason="host clang++ not available")

# (B, N, stride, K, mlp, dist): the three channel plans of the car config (64-64-128 with one wave per pooling window,
# 128-128-256, 256-256-512 with four waves per window and the 64 x 128 conv3 tiles), small enough for seconds on the host
CASES = [
    (2, 128, 3.5, 16, (64, 64, 128), 1.0),
    (3, 200, 2.5, 32, (64, 64, 128), 0.7),
    (2, 256, 2.0, 64, (128, 128, 256), 2.0),
    (2, 256, 4.0, 128, (256, 256, 512), 4.0),
]


@pytest.fixture()
def emu_native():
    """_native.lib() -> the host emulation of the library for the duration of one test."""
    from emu_fcn import emu_path
    from frustum_convnet_amd import _native
    saved = (_native.LIB_PATH, _native._lib)
    _native.LIB_PATH, _native._lib = emu_path(), None
    try:
        yield _native.lib()
    finally:
        _native.LIB_PATH, _native._lib = saved


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_N%d_K%d_C%d" % (c[0], c[1], c[3], c[4][2]))
def test_emulated_stages(emu_native, case):
    import gpu_stage_check as gsc
    res = gsc.run_stages(*case, verbose=False, emu=True)
    bad = gsc.check(res)
    assert not bad, bad
