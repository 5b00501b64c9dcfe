"""Builds libfcn_hip.so (the gfx950 HIP kernels + C-ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot.  One architecture, one code path: no fallbacks.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["grouping.hip", "group_compact.hip", "pointnet_fwd.hip", "pointnet_bwd.hip", "loss_tail.hip", "fcn_net.hip", "optim.hip", "inputs.hip", "box_iou.hip"]
HEADERS = ["fcn_common.h", "gemm_tile.h", "box_iou.h", os.path.join("..", "..", "include", "fcn_hip.h")]
LIB = os.path.join(HERE, "libfcn_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=off"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[fcn build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
