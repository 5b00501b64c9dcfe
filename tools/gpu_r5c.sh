#!/bin/bash
# Round 5, session c: K-groups inside the weight-gradient workgroups (G x 4 waves, in-LDS meet) x pre-encoded operands.
# libs: product (G = 2 plain / 4 PRE), kg1 (1 / 1 = the four-wave kernels), kg22 (2 / 2); FCN_PN_PRE = 1 / 0 on each.
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in libfcn_hip.so libfcn_hip_kg1.so libfcn_hip_kg22.so; do
  for v in 0 1; do
    echo "== pn_micro $lib FCN_PN_PRE=$v"; FCN_LIB_NAME=$lib FCN_PN_PRE=$v timeout 200 python tools/pn_micro.py 20 2>&1 | grep "scale [34]" | tee -a $O/r05_c_pn_micro.txt
  done
done
for i in 1 2; do
  for lib in libfcn_hip.so libfcn_hip_kg1.so libfcn_hip_kg22.so; do
    for v in 0 1; do
      FCN_LIB_NAME=$lib FCN_PN_PRE=$v timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/r05_c_${lib}_$v_$i.json 2> $O/r05_c_err.txt
      echo "$lib PRE=$v $i: $(python -c "import json,sys; d=json.loads(open('$O/r05_c_${lib}_$v_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $O/r05_c_bench.txt
    done
  done
done
