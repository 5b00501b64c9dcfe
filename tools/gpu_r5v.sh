#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
ENVVAR=FCN_TOPO VALUES="0 2" ROUNDS=2 TAG=r05_v_topo bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_v_ab.txt
unset FCN_TOPO
ENVVAR=FCN_POOL_KEYS VALUES="0 1" ROUNDS=2 TAG=r05_v_keys bash tools/gpu_ab_env.sh 2>&1 | tee -a $O/r05_v_ab.txt
