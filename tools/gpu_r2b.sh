#!/bin/bash
# Round-2 GPU session B: new kernels (box IoU / decode / NMS, IoU metrics in the loss tail, fused front), bf16-mode test,
# whole suite, bench with the fused front on / off.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_group_compact.py -m gpu -q --timeout 600 -x > $O/pytest_new.txt 2>&1; echo "rc=$?"; tail -40 $O/pytest_new.txt
echo "== new tests (rest, no -x)"; timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_group_compact.py -m gpu -q --timeout 600 > $O/pytest_new_all.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_new_all.txt | head -20
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_all.txt 2>&1; echo "rc=$?"; tail -15 $O/pytest_all.txt; grep -E "bf16 mode|IoU metrics|iou pair" $O/pytest_all.txt
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_box.py -m gpu -q -s -k "bf16 or iou" > $O/pytest_s.txt 2>&1; grep -E "bf16 mode|IoU metrics|iou pair" $O/pytest_s.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "rc=$?"; tail -2 $O/smoke.txt
for ff in 1 0; do
  echo "== bench fused_front=$ff"; FCN_FUSED_FRONT=$ff timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_ff$ff.txt 2> $O/bench_ff$ff.err; echo "rc=$?"; tail -1 $O/bench_ff$ff.txt | cut -c1-300; tail -3 $O/bench_ff$ff.err
done
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_b.txt 2>&1; tail -10 $O/phase_b.txt
echo "== rocprof"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench.txt 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats_b.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do head -4000 $f > $O/kernel_trace_b.csv; done
head -12 $O/kernel_stats_b.csv | cut -c1-160
