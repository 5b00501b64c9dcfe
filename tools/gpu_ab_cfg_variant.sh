#!/bin/bash
# product library against ONE tuning build over several configurations (CFGS), alternating, PAIRS pairs each
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; V=${VARIANT:?}; N=${PAIRS:-2}
for c in ${CFGS:-car people refine sunrgbd}; do
  for i in $(seq 1 $N); do
    for lib in prod $V; do
      if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
      timeout 120 python bench.py --cfg $c --no-cpu-baseline --no-roofline --no-configs --min-time 1.2 > $O/abc_${c}_${lib}_$i.json 2> $O/abc_${c}_${lib}_$i.err
      echo "$c $lib $i: $(python -c "import json,sys; d=json.loads(open('$O/abc_${c}_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
    done
  done
done
