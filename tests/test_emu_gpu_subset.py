"""The fast part of the -m gpu suite, run on the CPU: the SAME test functions (grouping, rotated-box kernels, input
construction, the fused front, and the whole model against the golden vectors of the reference for two configurations, every parameter gradient against the fp64 oracle) with
the package's GPU-only Python layer pointed at the host emulation of the kernels (tests/emu_shim.py + tests/host_harness).
The hardware run of these tests stays the parity gate; this tier catches index / layout / reduction mistakes -- in the kernels
and in the host code that drives them -- without a GPU.  (FCN_EMULATE=1 python -m pytest tests -m gpu -k ... runs any other
GPU test the same way; whole-model cases take about a minute each, hipGraph tests cannot be emulated.)"""
import importlib
import os
import shutil

import pytest

CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which(CLANG)), reason="host clang++ not available")

CASES = [
    ("test_gpu_grouping", "test_testpy_scenario", ()),
    ("test_gpu_grouping", "test_golden_full", ("car_b4_n512",)),
    ("test_gpu_grouping", "test_golden_full", ("car_b4_n512_uniform",)),
    ("test_gpu_grouping", "test_golden_full", ("refine_b4_n512",)),
    ("test_gpu_grouping", "test_golden_sha", ("car_b32_n1024",)),
    ("test_gpu_grouping", "test_golden_sha", ("people_b2_n512",)),
    ("test_gpu_grouping", "test_ragged_shapes", (1, 1, 1, 1, 0.5)),
    ("test_gpu_grouping", "test_ragged_shapes", (2, 65, 17, 64, 0.3)),
    ("test_gpu_grouping", "test_ragged_shapes", (3, 1000, 33, 200, 0.05)),
    ("test_gpu_grouping", "test_ragged_shapes", (1, 20000, 9, 16, 0.01)),
    ("test_gpu_grouping", "test_boundary_is_strict_and_fp32", ()),
    ("test_gpu_grouping", "test_kernel_native_layout_matches", ()),
    ("test_gpu_box", "test_iou_pair_matches_golden", ()),
    ("test_gpu_box", "test_rotate_nms_matches_reference_keep_lists", ()),
    ("test_gpu_box", "test_decode_matches_oracle", (3, 64)),
    ("test_gpu_box", "test_decode_matches_oracle", (10, 128)),
    ("test_gpu_inputs", "test_golden_batch_with_recorded_draws", ()),
    ("test_gpu_inputs", "test_same_numpy_seed_reproduces_the_reference_batch", ()),
    ("test_gpu_inputs", "test_against_oracle_without_augmentation_and_nearest_fallback", (True, False)),
    ("test_gpu_inputs", "test_against_oracle_without_augmentation_and_nearest_fallback", (False, True)),
    ("test_gpu_inputs", "test_ragged_and_odd_sizes_against_oracle", (3, 257, False)),
    ("test_gpu_inputs", "test_refine_builder_matches_reference_batch", ()),
    ("test_gpu_inputs", "test_sunrgbd_builder_matches_reference_batch", ()),
    ("test_gpu_group_compact", "test_group_compact_matches_oracle_and_unfused", (4, 512, (0.25, 0.5, 1.0, 2.0), "car")),
    ("test_gpu_model", "test_train_eval_parity", ("refine_b4_n512",)),
    ("test_gpu_model", "test_train_eval_parity", ("people_b2_n512",)),
    ("test_gpu_model", "test_gradients_vs_fp64_oracle", ()),
]


@pytest.mark.parametrize("mod,fn,args", CASES, ids=["%s-%s" % (c[1], "_".join(str(a) for a in c[2]))[:70] for c in CASES])
def test_gpu_test_under_emulation(mod, fn, args):
    from emu_shim import emulated_gpu
    m = importlib.import_module(mod)
    with emulated_gpu():
        getattr(m, fn)(*args)
