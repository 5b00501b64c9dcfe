#!/bin/bash
# PointNet backward GEMMs after the probe-guided fixes: parity, probe, A/B against the previous build on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_model.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -2
FCN_LIB_NAME=libfcn_hip_pnprobe.so timeout 300 python tools/pn_probe.py car 2>&1 | grep "grad" | cut -c1-330
run() { n=$1; shift
  env "$@" timeout 400 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_w_$n.txt 2> $O/bench_w_$n.err; echo "== $n rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_w_$n.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"]:
    if r["entry"].startswith("fcn_pn_"): print("  %-46s %.4f ms" % (r["entry"], r["ms_per_step"]))
PY
}
run prev FCN_LIB_NAME=libfcn_hip_prev.so
run new FCN_X=0
run prev2 FCN_LIB_NAME=libfcn_hip_prev.so
run new2 FCN_X=0
