#!/bin/bash
# TEST INFRASTRUCTURE: the kernels' host emulation (tests/host_harness) built with AddressSanitizer and every emulated GPU test run
# under it -- heap out-of-bounds / use-after-free of ANY kernel on ANY tensor aborts the run (tensors below 4 KB included, which the
# guard-page interposer of tests/host_harness/guard does not see).  CPU only, ~20 minutes with 10 workers.  Leaves the ASan build in
# tests/host_harness/_build/libfcn_emu.so: run `python tests/host_harness/build_emu.py --force` afterwards.
set -e
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests/host_harness")
import build_emu as b
src = b._stage_sources()
cmd = [b.CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-mf16c", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-w",
       "-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-DFCN_BWD_G4_ROWS=1", "-I", os.path.join(b.HERE, "hip_emu")]
subprocess.check_call(cmd + [os.path.join(src, s) for s in b.SOURCES] + ["-o", b.OUT])
PY
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:verify_asan_link_order=0 LD_PRELOAD=$RT \
    python -m pytest tests/test_emu_gpu_subset.py tests/test_emu_misc.py -q -n ${WORKERS:-10} -p no:cacheprovider
