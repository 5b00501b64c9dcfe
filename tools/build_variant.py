"""Tuning builds of the library (never the product): python tools/build_variant.py <name> [-DFLAG=V ...] writes
frustum_convnet_amd/libfcn_hip_<name>.so; select it with FCN_LIB_NAME=libfcn_hip_<name>.so (frustum_convnet_amd/_native.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frustum_convnet_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
print(b.build(lib=os.path.join(b.HERE, "libfcn_hip_%s.so" % name), extra_flags=extra, verbose=False))
