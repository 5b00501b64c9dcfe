#!/bin/bash
# Round-2 GPU session C: XCD-aware FCN tile order, two-launch fused front, IoU metrics on a side stream; FCN forward tile sweep
# (variant libraries through FCN_LIB_NAME); the rewritten bench (kernel table, cpu baseline variants, cfgs).
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_all.txt 2>&1; echo "rc=$?"; tail -12 $O/pytest_all.txt | cut -c1-200
echo "== bench default (full)"; timeout 900 python bench.py --steps 200 --warmup 30 > $O/bench_full.txt 2> $O/bench_full.err; echo "rc=$?"; tail -1 $O/bench_full.txt | cut -c1-600; tail -3 $O/bench_full.err
for ff in 0; do
  echo "== bench fused_front=$ff"; FCN_FUSED_FRONT=$ff timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_ff$ff.txt 2> $O/bench_ff$ff.err; echo "rc=$?"; tail -1 $O/bench_ff$ff.txt | cut -c1-260
done
for v in ft142 ft242 ft222 ft122; do
  echo "== variant $v"; FCN_LIB_NAME=libfcn_hip_$v.so timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "train_eval_parity and car_b4_n512" > $O/pytest_$v.txt 2>&1; echo "parity rc=$?"
  FCN_LIB_NAME=libfcn_hip_$v.so timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_$v.txt 2> $O/bench_$v.err; echo "rc=$?"; tail -1 $O/bench_$v.txt | cut -c1-260
done
for c in people refine; do
  echo "== bench cfg $c"; timeout 600 python bench.py --cfg $c --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_$c.txt 2> $O/bench_$c.err; echo "rc=$?"; tail -1 $O/bench_$c.txt | cut -c1-400; tail -2 $O/bench_$c.err
done
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_c.txt 2>&1; tail -10 $O/phase_c.txt
echo "== rocprof"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench.txt 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats_c.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do tail -2500 $f > $O/kernel_trace_c_tail.csv; head -1 $f > $O/kernel_trace_c_head.csv; done
head -8 $O/kernel_stats_c.csv | cut -c1-160
