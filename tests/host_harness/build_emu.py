"""TEST INFRASTRUCTURE ONLY.  Compiles kernel sources of frustum_convnet_amd/csrc UNMODIFIED for the host against the HIP
stand-in of tests/host_harness/hip_emu (clang++ -x c++): python tests/host_harness/build_emu.py -> _build/libfcn_emu.so.
Used by tests/test_emu_fcn.py to run the FCN kernels' exact index arithmetic / LDS choreography on the CPU against the oracle."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "frustum_convnet_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libfcn_emu.so")
SOURCES = ["fcn_net.hip", "grouping.hip", "group_compact.hip", "pointnet_fwd.hip", "pointnet_bwd.hip", "loss_tail.hip", "optim.hip", "inputs.hip", "box_iou.hip"]
CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False, extra=()):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip_emu", "hip", "hip_runtime.h"), __file__]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-g0", "-mf16c", "-fPIC", "-shared", "-ffp-contract=off", "-w",
           "-I", os.path.join(HERE, "hip_emu")] + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith("-D")]))
