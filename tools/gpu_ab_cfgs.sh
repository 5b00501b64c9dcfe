#!/bin/bash
# One-box A/B of a run-time switch over every shipped configuration: ENVVAR=FCN_PN_MID VALUES="0 auto" CFGS="car people refine sunrgbd"
mkdir -p gpurun_out; O=gpurun_out
for c in ${CFGS:-car people refine sunrgbd}; do
  for i in 1 2; do
    for v in $VALUES; do
      export $ENVVAR=$v
      r=$(timeout 120 python bench.py --cfg $c --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 2>$O/abc_${c}_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
      echo "$c $ENVVAR=$v $i: $r"
    done
  done
done
