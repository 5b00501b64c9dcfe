#!/bin/bash
# Round 5, session e: which operands to pre-encode -- FCN_PN_PRE = 0 (none) / 1 (dy only: no extra activation images) / 3 (all)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_pointnet.py -x -q -m gpu -k preencoded 2>&1 | tail -3
ENVVAR=FCN_PN_PRE VALUES="0 1 3" ROUNDS=3 TAG=r05_e_pre bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_e_ab.txt
for v in 0 1 3; do echo "== pn_micro FCN_PN_PRE=$v"; FCN_PN_PRE=$v timeout 200 python tools/pn_micro.py 20 2>&1 | grep "scale [34]" | tee -a $O/r05_e_pn_micro.txt; done
for v in 0 1; do echo "== stamps FCN_PN_PRE=$v"; FCN_PN_PRE=$v timeout 120 python tools/pn_bwd_stamps.py 2>&1 | tail -6 | tee $O/r05_e_pn_bwd_stamps_$v.txt; done
