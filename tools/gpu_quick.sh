#!/bin/bash
# Quick GPU check: model / FCN parity tests, bench (no CPU baseline), phase stamps.  QUICK_TESTS overrides the test selection.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; T=${QUICK_TAG:-q}
echo "== pytest"; timeout 900 python -m pytest ${QUICK_TESTS:-tests/test_gpu_model.py tests/test_gpu_train_state.py tests/test_gpu_properties.py} -m gpu -q --timeout 600 > $O/pytest_$T.txt 2>&1; echo "rc=$?"; tail -4 $O/pytest_$T.txt | cut -c1-200
echo "== bench"; timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_$T.txt 2> $O/bench_$T.err; echo "rc=$?"; tail -1 $O/bench_$T.txt | cut -c1-260; tail -2 $O/bench_$T.err
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_$T.txt 2>&1; tail -9 $O/phase_$T.txt
