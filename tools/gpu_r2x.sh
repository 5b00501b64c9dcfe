#!/bin/bash
export TMPDIR=/tmp
FCN_LIB_NAME=libfcn_hip_pnprobe.so timeout 300 python tools/pn_probe.py car 2>&1 | grep -E "grad" | cut -c1-330
