#!/bin/bash
# Round-2 GPU session A: split-MFMA bring-up.  Parity tests at the default (split) precision, the same subset in exact-f32
# mode (the staging layouts changed for every mode), bench at the three precisions, kernel stats of the split step.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== stage check split"; timeout 600 python tests/gpu_stage_check.py > $O/stage_split.txt 2>&1; echo "rc=$?"; grep -E "FAIL|Error|->|max" $O/stage_split.txt | head -40
echo "== stage check f32"; FCN_PRECISION=f32 timeout 600 python tests/gpu_stage_check.py > $O/stage_f32.txt 2>&1; echo "rc=$?"; grep -E "FAIL|Error|->|max" $O/stage_f32.txt | head -40
echo "== pytest gpu (split)"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_split.txt 2>&1; echo "rc=$?"; tail -40 $O/pytest_split.txt
echo "== pytest gpu (f32 subset)"; FCN_PRECISION=f32 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pointnet.py tests/test_gpu_train_state.py -m gpu -q --timeout 600 > $O/pytest_f32.txt 2>&1; echo "rc=$?"; tail -15 $O/pytest_f32.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "rc=$?"; tail -2 $O/smoke.txt
for p in split f32 bf16; do
  echo "== bench $p"; timeout 600 python bench.py --steps 300 --warmup 30 --precision $p --no-cpu-baseline > $O/bench_$p.txt 2> $O/bench_$p.err; echo "rc=$?"; tail -1 $O/bench_$p.txt | cut -c1-400; tail -3 $O/bench_$p.err
done
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/phase_split.txt 2>&1; tail -12 $O/phase_split.txt
echo "== rocprof split"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench.txt 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats_split.csv; done
head -40 $O/kernel_stats_split.csv | cut -c1-200
