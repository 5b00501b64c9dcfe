#!/bin/bash
# One replayed step under the rocprofv3 kernel trace, launch by launch (tools/trace_summary.py ... all): where the critical path's
# gaps are.  TAG names the output (gpurun_out/${TAG}_timeline.txt); extra environment (FCN_* switches) is inherited.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${TAG:-trace}
cd /tmp; rm -rf /tmp/prof_$T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$T -o b -- python $R/bench.py --steps 12 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs $BENCH_ARGS > $O/${T}_trace_bench.txt 2> $O/${T}_trace_bench.err
cd $R; for f in $(find /tmp/prof_$T -name "*kernel_trace*.csv"); do (head -1 $f; tail -1200 $f) > $O/${T}_kernel_trace.csv; done
python tools/trace_summary.py $O/${T}_kernel_trace.csv 12 all > $O/${T}_timeline.txt 2>&1
head -3 $O/${T}_timeline.txt; tail -1 $O/${T}_trace_bench.txt | cut -c1-200
