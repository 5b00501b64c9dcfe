#!/bin/bash
# Round 5, session i: is a rank's step of the N > 1 form (three graphs, Adam at the head, prefetch, two workspace sets) the N = 1 step?
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
ENVVAR=FCN_BENCH_SPLIT_STEP VALUES="0 1" ROUNDS=3 TAG=r05_i_split bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_i_ab.txt
python -c "
import json
for v in (0,1):
    d=json.loads(open('$O/r05_i_split_%d_1.json'%v).read().strip().split(chr(10))[-1]); print(v, d['final_loss'], d['config']['launch'])"
