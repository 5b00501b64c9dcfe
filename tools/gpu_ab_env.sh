#!/bin/bash
# One-box A/B of a RUN-TIME switch of the product library: ENVVAR=FCN_POOL_KEYS VALUES="1 0" (first value = the default).
# ROUNDS alternating rounds of short bench runs, then (PROF=1) per-kernel averages of each setting from a rocprofv3 kernel
# trace and (PHASES=1) the phase stamps of each.
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${TAG:-abe}
for i in $(seq 1 ${ROUNDS:-3}); do
  for v in $VALUES; do
    export $ENVVAR=$v
    timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 $BENCH_ARGS > $O/${T}_${v}_$i.json 2> $O/${T}_${v}_$i.err
    echo "$ENVVAR=$v $i: $(python -c "import json,sys; d=json.loads(open('$O/${T}_${v}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
if [ -n "$PHASES" ]; then
  for v in $VALUES; do export $ENVVAR=$v; echo "-- phases $ENVVAR=$v"; timeout 120 python tools/phase_stamps.py 2>&1 | tail -9 | tee $O/${T}_phases_$v.txt; done
fi
if [ -n "$PROF" ]; then
  for v in $VALUES; do
    export $ENVVAR=$v; cd /tmp; rm -rf /tmp/prof_$v
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o b -- python $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > /dev/null 2>&1
    cd $R; for f in $(find /tmp/prof_$v -name "*kernel_stats*.csv"); do cp $f $O/${T}_${v}_kernel_stats.csv; done
  done
  python tools/kernel_compare.py $(for v in $VALUES; do echo $O/${T}_${v}_kernel_stats.csv; done) | head -45
fi
