#!/bin/bash
# alternating bench pairs of the product library and ONE tuning build (VARIANT), N pairs (default 4), optional parity subset first
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; V=${VARIANT:?}; N=${PAIRS:-4}
if [ -n "$TESTS" ]; then
  FCN_LIB_NAME=libfcn_hip_$V.so timeout 400 python -m pytest $TESTS -x -q -m gpu > $O/ab2_${V}_pytest.txt 2>&1; echo "parity rc=$?"; tail -2 $O/ab2_${V}_pytest.txt
fi
for i in $(seq 1 $N); do
  for lib in prod $V; do
    if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
    timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/ab2_${lib}_$i.json 2> $O/ab2_${lib}_$i.err
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/ab2_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
