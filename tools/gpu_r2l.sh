#!/bin/bash
# Round-2 GPU session L: deeper operand prefetch in the PointNet forward GEMM (variants pndb1: two register + two LDS stages,
# pndb2: two register stages) against the product build; per-entry times from bench's live kernel table.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_l_$n.txt 2> $O/bench_l_$n.err; echo "== $n rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_l_$n.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"]:
    if r["entry"].startswith("fcn_pn_forward"): print("  %-46s %.4f ms" % (r["entry"], r["ms_per_step"]))
PY
}
run base FCN_X=0
run pndb1 FCN_LIB_NAME=libfcn_hip_pndb1.so
run pndb2 FCN_LIB_NAME=libfcn_hip_pndb2.so
run base2 FCN_X=0
timeout 300 python -m pytest tests/test_gpu_pointnet.py -m gpu -q 2>&1 | tail -2
FCN_LIB_NAME=libfcn_hip_pndb1.so timeout 300 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_model.py -m gpu -q -k "pointnet or car_b32" 2>&1 | tail -2
FCN_LIB_NAME=libfcn_hip_pndb2.so timeout 300 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_model.py -m gpu -q -k "pointnet or car_b32" 2>&1 | tail -2
