#!/bin/bash
# Round 5, session x: re-sweep of run-time switches that earlier rounds settled (their balance may have moved)
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
run() { # name env...
  local name=$1; shift
  for i in 1 2; do
    echo "$name $i: $(env "$@" timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/r05_x_sweep.txt
  done
}
run base A=1
run store_dy3_0 FCN_STORE_DY3=0
run mid_0 FCN_PN_MID=0
run mid_1 FCN_PN_MID=1
run spg4 FCN_STEPS_PER_GRAPH=4
run pack_behind FCN_PACK_ORDER=behind
run pack_beside FCN_PACK_ORDER=beside
run iou_early FCN_IOU_JOIN=early
run pf_bwd FCN_PF_POINT=bwd
run base2 A=1
