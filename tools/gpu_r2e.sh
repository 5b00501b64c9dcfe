#!/bin/bash
# Round-2 GPU session E: early-issued FCN loads, parallel moment fetch in the fused front, wide-lane pool, refine inputs;
# timing-only experiment builds (operands stored without encoding: results wrong, bounds what pre-encoded operands could buy).
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_all.txt 2>&1; echo "rc=$?"; tail -6 $O/pytest_all.txt | cut -c1-200
echo "== bench default"; timeout 900 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_e.txt 2> $O/bench_e.err; echo "rc=$?"; tail -1 $O/bench_e.txt | cut -c1-300; tail -3 $O/bench_e.err
for v in exp1 exp3; do
  echo "== timing experiment $v"; FCN_LIB_NAME=libfcn_hip_$v.so timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_$v.txt 2> $O/bench_$v.err; echo "rc=$?"; tail -1 $O/bench_$v.txt | cut -c1-260
  FCN_LIB_NAME=libfcn_hip_$v.so timeout 300 python tools/phase_stamps.py > $O/phase_$v.txt 2>&1; tail -9 $O/phase_$v.txt
done
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_e.txt 2>&1; tail -10 $O/phase_e.txt
echo "== rocprof car"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench_e.txt 2> $GRAFT_REPO_ROOT/$O/prof_e.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats_e.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do (head -1 $f; tail -1200 $f) > $O/kernel_trace_e.csv; done
head -6 $O/kernel_stats_e.csv | cut -c1-160
