#!/bin/bash
# Round-2 GPU session I: the 5-level SUN-RGBD variant (det_base_sunrgbd) -- parity suite, bench of the car config (regression
# check of the generalised FCN plan) and of the sunrgbd config.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest gpu (sunrgbd + model first)"; timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -s > $O/pytest_model.txt 2>&1; echo "rc=$?"; grep -E "sunrgbd|passed|failed|Error" $O/pytest_model.txt | cut -c1-220 | tail -12
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_all.txt 2>&1; echo "rc=$?"; tail -6 $O/pytest_all.txt | cut -c1-200
echo "== bench car"; timeout 900 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_i.txt 2> $O/bench_i.err; echo "rc=$?"; tail -1 $O/bench_i.txt | cut -c1-300; tail -3 $O/bench_i.err
echo "== bench sunrgbd"; timeout 900 python bench.py --cfg sunrgbd --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_i_sunrgbd.txt 2> $O/bench_i_sunrgbd.err; echo "rc=$?"; tail -1 $O/bench_i_sunrgbd.txt | cut -c1-300; tail -3 $O/bench_i_sunrgbd.err
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_i.txt 2>&1; tail -10 $O/phase_i.txt
