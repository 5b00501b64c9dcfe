#!/bin/bash
# Round-2 GPU session O: BatchNorm finalisation / BN-backward coefficients / layer-1 gradients by the last workgroup of their
# producers (12 one-workgroup launches per step less) -- A/B against the previous build on one box, then the parity suite.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_o_$n.txt 2> $O/bench_o_$n.err; echo "== $n rc=$?"; tail -1 $O/bench_o_$n.txt | cut -c1-160
  env "$@" timeout 300 python tools/phase_stamps.py 2>&1 | grep -E "pointnet_fwd_done|backward_done|fcn_bwd_done"
}
run prev FCN_LIB_NAME=libfcn_hip_prev.so
run new FCN_X=0
run prev2 FCN_LIB_NAME=libfcn_hip_prev.so
run new2 FCN_X=0
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_o.txt 2>&1; echo "rc=$?"; tail -4 $O/pytest_o.txt | cut -c1-200
