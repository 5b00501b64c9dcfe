#!/bin/bash
# exact-fp32 mode, people full-size fixture: gradient norms of scale 1 with the product's weight-gradient split count and with 8x
# as many splits (8x shorter fp32 MFMA accumulation chains): does the chain length explain the mode's distance from fp64?
O=gpurun_out; mkdir -p $O
for lib in prod slots2k; do
  if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
  echo "== $lib"; timeout 300 python tools/grad_norm_check.py ${CASE:-people_b32_n1024} f32 2>&1 | grep "pointnet1\|Error\|error" | head -12
done | tee $O/f32_chain.txt
