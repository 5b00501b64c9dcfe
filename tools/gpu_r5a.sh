#!/bin/bash
# Round 5, session a: the pure-GEMM ceiling (tools/micro/pgemm), the PointNet GPU tests with the pre-encoded weight-gradient operands, and
# the one-box A/B FCN_PN_PRE = 1 / 0 (bench rounds, phase stamps, per-kernel averages, isolated kernels).
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== pgemm"; timeout 120 tools/micro/pgemm 2>&1 | tee $O/r05_a_pgemm.txt
echo "== pgemm dgrad shape (K=512, N=256)"; timeout 120 tools/micro/pgemm 36363 512 256 2>&1 | tee -a $O/r05_a_pgemm.txt
echo "== pytest pointnet"; timeout 600 python -m pytest tests/test_gpu_pointnet.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r05_a_pytest_pointnet.txt
echo "== A/B"; ENVVAR=FCN_PN_PRE VALUES="1 0" ROUNDS=3 PHASES=1 PROF=1 TAG=r05_a_pre bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_a_ab.txt
for v in 1 0; do echo "== pn_micro FCN_PN_PRE=$v"; FCN_PN_PRE=$v timeout 200 python tools/pn_micro.py 20 2>&1 | tail -14 | tee $O/r05_a_pn_micro_$v.txt; done
