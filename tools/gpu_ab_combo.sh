#!/bin/bash
# One-box comparison of several ENVIRONMENT COMBINATIONS of the product library: COMBOS="name:VAR=V,VAR2=V2 name2:..." (two alternating rounds)
mkdir -p gpurun_out; O=gpurun_out
run() { n=$1; shift
  r=$(env "$@" timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 $BENCH_ARGS 2>$O/combo_$n.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "$n: $r"; }
for i in $(seq 1 ${ROUNDS:-2}); do
  for c in $COMBOS; do
    n=${c%%:*}; e=${c#*:}
    run $n $(echo $e | tr ',' ' ')
  done
done
