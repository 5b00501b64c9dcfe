#!/bin/bash
# A/B on one box: pooling windows split across waves (new) vs one wave per window (prev); parity first.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_model.py tests/test_gpu_group_compact.py -m gpu -q 2>&1 | tail -3
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_q_$n.txt 2> $O/bench_q_$n.err; echo "== $n rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_q_$n.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"]:
    if r["entry"].startswith("fcn_pn_forward"): print("  %-46s %.4f ms" % (r["entry"], r["ms_per_step"]))
PY
  env "$@" timeout 300 python tools/phase_stamps.py 2>&1 | grep -E "pointnet_fwd_done"
}
run prev FCN_LIB_NAME=libfcn_hip_prev.so
run new FCN_X=0
run prev2 FCN_LIB_NAME=libfcn_hip_prev.so
run new2 FCN_X=0
FCN_X=0 timeout 300 python bench.py --cfg sunrgbd --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
FCN_LIB_NAME=libfcn_hip_prev.so timeout 300 python bench.py --cfg sunrgbd --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
