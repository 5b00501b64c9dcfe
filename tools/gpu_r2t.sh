#!/bin/bash
# A/B on one box: prev (HEAD) vs new (FCN reduce role with 4 elements per thread) vs occ (new + higher occupancy targets for
# the widest PointNet conv3 / the 64 x 128 dgrad tiles); parity of the touched paths first.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_state.py -m gpu -q 2>&1 | tail -2
FCN_LIB_NAME=libfcn_hip_occ.so timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_pointnet.py -m gpu -q 2>&1 | tail -2
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_t_$n.txt 2> $O/bench_t_$n.err; echo "== $n rc=$?"; tail -1 $O/bench_t_$n.txt | cut -c1-150
  env "$@" timeout 300 python tools/phase_stamps.py 2>&1 | grep -E "pointnet_fwd_done|backward_done|fcn_bwd_done"
}
run prev FCN_LIB_NAME=libfcn_hip_prev.so
run new FCN_X=0
run occ FCN_LIB_NAME=libfcn_hip_occ.so
run prev2 FCN_LIB_NAME=libfcn_hip_prev.so
run new2 FCN_X=0
run occ2 FCN_LIB_NAME=libfcn_hip_occ.so
