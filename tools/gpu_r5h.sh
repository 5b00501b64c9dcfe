#!/bin/bash
# Round 5, session h: N = 1 line, then the N = 2 step on ONE GPU over gloo with and without its collectives (FCN_SKIP_COMM=1: is a rank's
# step the N = 1 step?), then the dist / model / pointnet GPU tests
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
  timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/r05_h_n1_$i.json 2> $O/r05_h_n1.err
  echo "N=1 $i: $(python -c "import json; d=json.loads(open('$O/r05_h_n1_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])" 2>&1 | tail -1)"
  for sk in 1 0; do
    HSA_ENABLE_IPC_MODE_LEGACY=0 FCN_SKIP_COMM=$sk FCN_BENCH_BACKEND=gloo FCN_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 2 --steps 20 --warmup 6 --min-time 1.0 > $O/r05_h_n2_skip${sk}_$i.json 2> $O/r05_h_n2_skip${sk}.err
    echo "N=2 one GPU, gloo, skip_comm=$sk $i: $(python -c "import json; d=json.loads(open('$O/r05_h_n2_skip${sk}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:90])" 2>&1 | tail -1)"; tail -2 $O/r05_h_n2_skip${sk}.err | cut -c1-300
  done
done
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_pointnet.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4 | tee $O/r05_h_pytest.txt
