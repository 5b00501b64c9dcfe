#!/bin/bash
# Round 5, session g: tiles per workgroup in the FCN forward (T): product (2 from 768 tiles), ft0 (always 1 = round 4), ft4 (4 from 1000 tiles),
# ft2b (2 from 500 tiles: every layer)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused_convnet or train_eval_parity" 2>&1 | tail -3
for i in 1 2 3; do
  for lib in libfcn_hip.so libfcn_hip_ft0.so libfcn_hip_ft4.so libfcn_hip_ft2b.so; do
    FCN_LIB_NAME=$lib timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/r05_g_${lib}_$i.json 2> $O/r05_g_err.txt
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/r05_g_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $O/r05_g_bench.txt
  done
done
for lib in libfcn_hip.so libfcn_hip_ft0.so libfcn_hip_ft4.so; do echo "-- phases $lib"; FCN_LIB_NAME=$lib timeout 120 python tools/phase_stamps.py 2>&1 | tail -9 | tee $O/r05_g_phases_$lib.txt; done
