#!/bin/bash
# Round 5, session m: the tail launch (both reduces + layer-1 finalisation in one launch, reduces off the path between the GEMMs)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pointnet.py -x -q -m gpu 2>&1 | tail -3
ENVVAR=FCN_PN_TAIL VALUES="1 0" ROUNDS=4 TAG=r05_m_tail bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_m_ab.txt
for v in 1 0; do echo "== stamps FCN_PN_TAIL=$v"; FCN_PN_TAIL=$v timeout 120 python tools/pn_bwd_stamps.py 2>&1 | tail -6 | tee $O/r05_m_pn_bwd_stamps_$v.txt; done
