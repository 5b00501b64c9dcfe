#!/bin/bash
# Round-2 GPU session D: loss tail with LDS rows, unit-gradient seed, XCD order in the PointNet GEMMs; wide forward tiles
# variant; kernel traces of the car and refine steps.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_all.txt 2>&1; echo "rc=$?"; tail -6 $O/pytest_all.txt | cut -c1-200
echo "== bench default (full)"; timeout 900 python bench.py --steps 200 --warmup 30 > $O/bench_full.txt 2> $O/bench_full.err; echo "rc=$?"; tail -1 $O/bench_full.txt | cut -c1-400; tail -3 $O/bench_full.err
echo "== variant wide"; FCN_LIB_NAME=libfcn_hip_wide.so timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "train_eval_parity and car_b4_n512" > $O/pytest_wide.txt 2>&1; echo "parity rc=$?"
FCN_LIB_NAME=libfcn_hip_wide.so timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_wide.txt 2> $O/bench_wide.err; echo "rc=$?"; tail -1 $O/bench_wide.txt | cut -c1-260
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_d.txt 2>&1; tail -10 $O/phase_d.txt
for c in car refine; do
echo "== rocprof $c"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --cfg $c --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench_$c.txt 2> $GRAFT_REPO_ROOT/$O/prof_$c.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats_d_$c.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do (head -1 $f; tail -1200 $f) > $O/kernel_trace_d_$c.csv; done
done
head -6 $O/kernel_stats_d_car.csv | cut -c1-160
