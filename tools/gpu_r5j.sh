#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
export FCN_BENCH_SPLIT_STEP=1
ENVVAR=FCN_BENCH_PIECES VALUES="3 2" ROUNDS=3 TAG=r05_j_pieces bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_j_ab.txt
unset FCN_BENCH_SPLIT_STEP
timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('N=1 form', d['value'], d['ms_per_step'])" | tee -a $O/r05_j_ab.txt
