#!/bin/bash
# Round-2 GPU session N: weight-gradient split count of the PointNet backward (partials written + re-read: 400 MB per step).
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_n_$n.txt 2> $O/bench_n_$n.err; echo "== $n rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_n_$n.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"]:
    if r["entry"].startswith("fcn_pn_backward2"): print("  %-46s %.4f ms" % (r["entry"], r["ms_per_step"]))
PY
  env "$@" timeout 300 python tools/phase_stamps.py 2>&1 | grep -E "backward_done|fcn_bwd_done"
}
run base FCN_X=0
run ws384 FCN_LIB_NAME=libfcn_hip_ws384.so
run ws256 FCN_LIB_NAME=libfcn_hip_ws256.so
run ws1536 FCN_LIB_NAME=libfcn_hip_ws1536.so
run base2 FCN_X=0
