"""TEST INFRASTRUCTURE ONLY.  Compiles the kernel sources of frustum_convnet_amd/csrc for the HOST against the HIP stand-in of
tests/host_harness/hip_emu (clang++ -x c++): python tests/host_harness/build_emu.py -> _build/libfcn_emu.so with the
library's whole C-ABI.  tests/test_emu_*.py run the kernels' exact index arithmetic / LDS choreography / reductions on the CPU
against the oracles with it.

The sources are compiled from a copy under _build/src with TWO mechanical substitutions (PATCHES below), both about things a
sequential host execution cannot express and neither touching arithmetic:
  * `extern __shared__ ... T name[];` (dynamic LDS) -> a pointer to the per-launch buffer of the emulation;
  * the barrier-free K loop of the FCN forward hands LDS data between the lanes of ONE wave, which is ordered by the
    hardware's lockstep execution: the emulation runs lanes as coroutines and needs an explicit wave-level rendezvous there
    (`__builtin_amdgcn_wave_barrier()`, which the stand-in maps to one).
A source that already spells these out (FCN_DYN_LDS, wave barriers in place) is left as it is."""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "frustum_convnet_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
OUT = os.path.join(BUILD, "libfcn_emu.so")
SOURCES = ["fcn_net.hip", "grouping.hip", "group_compact.hip", "pointnet_fwd.hip", "pointnet_bwd.hip", "loss_tail.hip",
           "optim.hip", "inputs.hip", "box_iou.hip"]
CLANG = os.environ.get("FCN_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")

PATCHES = [
    (re.compile(r"extern\s+__shared__\s+__attribute__\(\(aligned\(16\)\)\)\s+([A-Za-z_ ]+?)\s+(\w+)\[\];"),
     r"\1 *\2 = (\1 *)emu::dyn_lds;"),
    (re.compile(r"if constexpr \(TG != 64\) __syncthreads\(\);(\s*\\)"),
     r"if constexpr (TG != 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();\1"),
]


def _stage_sources():
    dst = os.path.join(BUILD, "src", "frustum_convnet_amd", "csrc")
    inc = os.path.join(BUILD, "src", "include")
    os.makedirs(dst, exist_ok=True)
    os.makedirs(inc, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "fcn_hip.h"), inc)
    for f in os.listdir(CSRC):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        for pat, rep in PATCHES:
            text = pat.sub(rep, text)
        with open(os.path.join(dst, f), "w") as fh:
            fh.write(text)
    return dst


def build(force=False, extra=(), out=None):
    """out: another output path (a tuning build of the emulation: extra = its -D flags; always rebuilt when the sources changed)."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip_emu", "hip", "hip_runtime.h"), __file__,
                                                                os.path.join(ROOT, "include", "fcn_hip.h")]
    out = out or OUT
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    src = _stage_sources()
    # -DFCN_BWD_G4_ROWS=1: the FCN backward's FOUR-wave workgroups (csrc/fcn_net.hip cn_bwd_groups) for every shape -- on the GPU the
    # small test shapes run the eight-wave kernels and only the full-size fixtures the four-wave ones; here it is the other way round,
    # so both instantiations see small, ragged shapes somewhere
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-g0", "-mf16c", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-w",
           "-DFCN_BWD_G4_ROWS=1", "-I", os.path.join(HERE, "hip_emu")] + list(extra) + [os.path.join(src, s) for s in SOURCES] + ["-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith("-D")]))
