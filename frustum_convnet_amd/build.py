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
HEADERS = ["fcn_common.h", "fcn_tuning.h", "gemm_tile.h", "pn_pack.h", "box_iou.h", os.path.join("..", "..", "include", "fcn_hip.h")]
LIB = os.path.join(HERE, "libfcn_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=off"]


OBJ = os.path.join(CSRC, "_obj")


def source_hash(extra_flags=()):
    """sha256 over the kernel sources, headers and compile flags (hex, 16 characters): what libfcn_hip.so reports through
    fcn_build_hash() -- the library ships prebuilt to the GPU box, so smoke() and bench.py compare the two."""
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def _deps_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + [os.path.abspath(__file__)])


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, lib=None, extra_flags=()):
    """Compiles every source to an object (in parallel; only the ones older than their source / the headers) and links them.
    lib / extra_flags: tuning builds (tools/build_variant.py) -- objects of such a build are never cached."""
    variant = lib is not None or bool(extra_flags)
    lib = lib or LIB
    if not force and not variant and not _stale():
        return lib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    objdir = OBJ + ("_variant_%d" % os.getpid() if variant else "")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _deps_mtime()

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        if not force and not variant and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj
        cmd = [hipcc] + cflags + ["-c", path, "-o", obj]
        if verbose:
            print("[fcn build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the hash of what was compiled, linked in as a one-function host object
    hsrc, hobj = os.path.join(objdir, "build_hash.cpp"), os.path.join(objdir, "build_hash.o")
    with open(hsrc, "w") as f:
        f.write('extern "C" const char *fcn_build_hash(void) { return "%s"; }\n' % source_hash(extra_flags))
    import shutil
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if cxx:
        subprocess.check_call([cxx, "-O1", "-fPIC", "-c", hsrc, "-o", hobj])
    else:                                   # a ROCm box without a host compiler: hipcc compiles the one-line host object too
        subprocess.check_call([hipcc, "-x", "c++", "-O1", "-fPIC", "-c", hsrc, "-o", hobj])
    objs.append(hobj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print("[fcn build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if variant:
        shutil.rmtree(objdir, ignore_errors=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
