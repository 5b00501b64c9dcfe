#!/bin/bash
# Round 5, session f: intra-kernel stamps of the FCN backward roles and of the FCN forward (probe builds)
mkdir -p gpurun_out; O=gpurun_out
FCN_LIB_NAME=libfcn_hip_probe3.so timeout 200 python tools/fcn_probe_bwd.py 2>&1 | grep -v amdgpu.ids | tee $O/r05_f_fcn_probe_bwd.txt
FCN_LIB_NAME=libfcn_hip_probe1.so timeout 200 python tools/fcn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05_f_fcn_probe_fwd.txt
