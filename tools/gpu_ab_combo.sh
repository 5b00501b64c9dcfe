#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
run() { # name, env...
  n=$1; shift
  r=$(env "$@" timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 2>$O/combo_$n.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "$n: $r"
}
for i in 1 2; do
run base A=1
run mid140_280 FCN_PN_MID_L=140,280
run side0_midall FCN_PN_SIDE=0 FCN_PN_MID=1
run side0_mid35 FCN_PN_SIDE=0 FCN_PN_MID_L=35
run side0_mid35_140_280 FCN_PN_SIDE=0 FCN_PN_MID_L=35,140,280
run side0 FCN_PN_SIDE=0
done
FCN_PN_SIDE=0 FCN_PN_MID_L=35,140,280 python tools/pn_bwd_stamps.py 2>&1 | tail -6
