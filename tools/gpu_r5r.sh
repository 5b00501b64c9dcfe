#!/bin/bash
# Round 5, session r: the side stream's two weight-gradient GEMMs as ONE launch (product) against two launches (wgp0)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pointnet.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3 4; do
  for lib in libfcn_hip.so libfcn_hip_wgp0.so; do
    FCN_LIB_NAME=$lib timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/r05_r_${lib}_$i.json 2> $O/r05_r_err.txt
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/r05_r_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $O/r05_r_bench.txt
  done
done
for lib in libfcn_hip.so libfcn_hip_wgp0.so; do echo "== stamps $lib"; FCN_LIB_NAME=$lib timeout 120 python tools/pn_bwd_stamps.py 2>&1 | tail -6 | tee $O/r05_r_pn_bwd_stamps_$lib.txt; done
