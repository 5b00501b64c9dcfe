#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
ENVVAR=FCN_ADAM_LATE VALUES="0 1" ROUNDS=4 TAG=r05_u_late bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_u_ab.txt
