#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box: two ranks on GPU 0 with gloo as the transport (an RCCL communicator needs
# two devices).  Exercises measure()'s two-graph capture, the bucketed all-reduce between / after the replays and the rank-0 line.
mkdir -p gpurun_out
# (round 6: no launcher here -- `bench.py --gpus 2` starts its two ranks itself; a CPU transport takes the host-issued three-graph form)
HSA_ENABLE_IPC_MODE_LEGACY=0 FCN_BENCH_BACKEND=gloo FCN_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --min-time 0.3 > gpurun_out/n2_rehearsal.json 2> gpurun_out/n2_rehearsal.err
echo "rc=$?"; tail -1 gpurun_out/n2_rehearsal.json | cut -c1-600; tail -3 gpurun_out/n2_rehearsal.err | cut -c1-300
