#!/bin/bash
# round 5, session z3: second helping of the staging diet (SGPR-base loads + no row mask in the PointNet data gradient, vector
# coefficient reads in the FCN data gradient) against the previous commit's library (d1), one box
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== parity"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_pointnet.py -x -q -m gpu > $O/r5z3_pytest.txt 2>&1; echo "rc=$?"; tail -3 $O/r5z3_pytest.txt
VARIANT=d1 SKIP_TESTS=1 bash tools/gpu_ab_variant.sh
for i in 3 4; do
  for lib in prod d1; do
    if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
    timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/ab_${lib}_$i.json 2> $O/ab_${lib}_$i.err
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/ab_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
