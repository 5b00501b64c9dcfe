"""Tuning builds of the library (never the product): python tools/build_variant.py <name> [-DFLAG=V ...] writes
frustum_convnet_amd/libfcn_hip_<name>.so; select it with FCN_LIB_NAME=libfcn_hip_<name>.so (frustum_convnet_amd/_native.py)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frustum_convnet_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.HERE, "libfcn_hip_%s.so" % name)
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + extra + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", out]
subprocess.check_call(cmd)
print(out)
