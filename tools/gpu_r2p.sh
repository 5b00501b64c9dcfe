#!/bin/bash
# A/B on one box: previous build (separate finalisation launches) vs last-workgroup finalisation
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_p_$n.txt 2> $O/bench_p_$n.err; echo "== $n rc=$?"; tail -1 $O/bench_p_$n.txt | cut -c1-160; tail -2 $O/bench_p_$n.err
  env "$@" timeout 300 python tools/phase_stamps.py 2>&1 | grep -E "pointnet_fwd_done|backward_done|fcn_bwd_done"
}
run prev FCN_LIB_NAME=libfcn_hip_prev.so
run new FCN_X=0
run prev2 FCN_LIB_NAME=libfcn_hip_prev.so
run new2 FCN_X=0
