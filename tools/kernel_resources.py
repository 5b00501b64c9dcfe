"""Per-kernel register / scratch / LDS / occupancy table of libfcn_hip.so's sources (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py [source.hip ...]   (default: every source of the library)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frustum_convnet_amd import build as fb  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def main():
    srcs = sys.argv[1:] or fb.SOURCES
    for src in srcs:
        path = src if os.path.exists(src) else os.path.join(fb.CSRC, src)
        cmd = ["/opt/rocm/bin/hipcc"] + [f for f in fb.FLAGS if f not in ("-shared",)] + \
              [f for f in os.environ.get("FCN_EXTRA_FLAGS", "").split() if f] + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", path, "-o", "/dev/null"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        rows, cur = [], None
        for line in err.split("\n"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
        names = demangle([r["name"] for r in rows])
        print("== %s" % os.path.basename(path))
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
            print("  %-58s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occ %d" % (
                n[:58], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))


if __name__ == "__main__":
    main()
