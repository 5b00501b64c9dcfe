#!/bin/bash
# Round 5, session s: re-sweep of the weight-gradient split targets behind the tail launch (PointNet FCN_WG_SLOTS 256, FCN FCN_WG_TARGET 512)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
  for lib in libfcn_hip.so libfcn_hip_s384.so libfcn_hip_s192.so libfcn_hip_t384.so libfcn_hip_t768.so; do
    FCN_LIB_NAME=$lib timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.2 > $O/r05_s_${lib}_$i.json 2> $O/r05_s_err.txt
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/r05_s_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $O/r05_s_bench.txt
  done
done
