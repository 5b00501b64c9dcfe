#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
ENVVAR=FCN_ADAM_EARLY VALUES="-1 0 1 2" ROUNDS=3 TAG=r05_l_adam bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_l_ab.txt
