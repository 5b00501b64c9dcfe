#!/bin/bash
# One-box comparison of several tuning builds against the product library: VARIANTS="a b" (frustum_convnet_amd/libfcn_hip_<v>.so).
# Two alternating rounds of short bench runs, then (PROF=1) per-kernel averages of each from a rocprofv3 kernel trace.
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { if [ $1 = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$1.so; fi; }
for i in 1 2; do
  for lib in prod $VARIANTS; do
    run $lib
    timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 $BENCH_ARGS > $O/abm_${lib}_$i.json 2> $O/abm_${lib}_$i.err
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/abm_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
if [ -n "$PROF" ]; then
for lib in prod $VARIANTS; do
  run $lib; cd /tmp; rm -rf /tmp/prof_$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -o b -- python $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > /dev/null 2>&1
  cd $R; for f in $(find /tmp/prof_$lib -name "*kernel_stats*.csv"); do cp $f $O/abm_${lib}_kernel_stats.csv; done
done
python tools/kernel_compare.py $(for lib in prod $VARIANTS; do echo $O/abm_${lib}_kernel_stats.csv; done) | head -40
fi
