#!/bin/bash
# Round-2 GPU session K: where does the FCN backward launch time go (timing-only builds without wgrad / dgrad roles), and the
# stream-topology switches re-measured on the round-2 kernels.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { # name, env...
  n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline > $O/bench_k_$n.txt 2> $O/bench_k_$n.err; echo "== $n rc=$?"; tail -1 $O/bench_k_$n.txt | cut -c1-200
  env "$@" timeout 300 python tools/phase_stamps.py > $O/phase_k_$n.txt 2>&1; tail -9 $O/phase_k_$n.txt | head -8
}
run base FCN_X=0
run nowgrad FCN_LIB_NAME=libfcn_hip_nowgrad.so
run nodgrad FCN_LIB_NAME=libfcn_hip_nodgrad.so
run topo2 FCN_TOPO=2
run topo4 FCN_TOPO=4
run topo6 FCN_TOPO=6
run base2 FCN_X=0
