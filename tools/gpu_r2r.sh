#!/bin/bash
# Rehearsal of the N = 2 bench path (two graphs + overlapped bucketed all-reduce + Adam) with both ranks on the one GPU of
# this box over gloo, the loss-tail change's parity, and the 2-rank test file (skips without 2 GPUs).
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_box.py tests/test_gpu_train_state.py tests/test_gpu_dist.py -m gpu -q 2>&1 | tail -3
FCN_BENCH_BACKEND=gloo FCN_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_r_n2.txt 2> $O/bench_r_n2.err; echo "n2 rc=$?"; tail -1 $O/bench_r_n2.txt | cut -c1-400; tail -3 $O/bench_r_n2.err | cut -c1-200
FCN_BENCH_BACKEND=gloo FCN_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-overlap > $O/bench_r_n2no.txt 2> $O/bench_r_n2no.err; echo "n2 no-overlap rc=$?"; tail -1 $O/bench_r_n2no.txt | cut -c1-300
