#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
ENVVAR=FCN_POOL_KEYS VALUES="1 0" ROUNDS=4 PHASES=1 TAG=r05_w_keys bash tools/gpu_ab_env.sh 2>&1 | tee $O/r05_w_ab.txt
for c in people sunrgbd refine; do
  for v in 1 0; do
    echo "$c FCN_POOL_KEYS=$v: $(FCN_POOL_KEYS=$v timeout 120 python bench.py --cfg $c --no-cpu-baseline --no-roofline --no-configs --min-time 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/r05_w_ab.txt
  done
done
FCN_POOL_KEYS=1 timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/r05_w_pytest_keys.txt
