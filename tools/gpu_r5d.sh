#!/bin/bash
# Round 5, session d: PRE with conv2's weight gradient back on the side stream (main chain as short as without the images)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
ENVVAR=FCN_PN_PRE VALUES="1 0" ROUNDS=3 TAG=r05_d_pre bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_d_ab.txt
for v in 1 0; do echo "== stamps FCN_PN_PRE=$v"; FCN_PN_PRE=$v timeout 120 python tools/pn_bwd_stamps.py 2>&1 | tail -6 | tee $O/r05_d_pn_bwd_stamps_$v.txt; done
