#!/bin/bash
# One GPU-box session: stage diff table, -m gpu tests, smoke, bench, rocprof summary.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-30}
if [ -z "$SKIP_STAGE" ]; then echo "== stage check"; timeout 600 python tests/gpu_stage_check.py > gpurun_out/stage.txt 2>&1; echo "rc=$?"; grep -E "FAIL|Error|->" gpurun_out/stage.txt | head -20; fi
if [ -z "$SKIP_TESTS" ]; then echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/pytest_gpu.txt; fi
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.txt
echo "== bench"; timeout 900 python bench.py --steps $STEPS --warmup 5 $BENCH_ARGS > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "rc=$?"; tail -2 gpurun_out/bench.txt; tail -5 gpurun_out/bench.err
if [ -n "$PROF" ]; then
  echo "== rocprof"; cd /tmp; rm -rf /tmp/prof; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err; echo "rc=$?"
  cd $GRAFT_REPO_ROOT; find /tmp/prof -type f | head -20; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/kernel_stats.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do cp $f gpurun_out/kernel_trace.csv; done
  head -30 gpurun_out/kernel_stats.csv; tail -1 gpurun_out/prof_bench.txt
fi
if [ -n "$PHASE" ]; then echo "== phase timing"; timeout 600 python tools/phase_timing.py > gpurun_out/phase.txt 2>&1; echo "rc=$?"; cat gpurun_out/phase.txt | grep -v Warning | tail -20; fi
