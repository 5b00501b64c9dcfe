#!/bin/bash
# Round 5, session b: where the PRE step loses -- timelines of one replayed step, the PointNet backward's un-traced stamps, and per-kernel
# times of the widest scale's backward in isolation (eager, one stream) for FCN_PN_PRE = 1 / 0.
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 1 0; do
  export FCN_PN_PRE=$v
  echo "== trace FCN_PN_PRE=$v"; TAG=r05_b_pre$v bash tools/gpu_trace.sh
  echo "== stamps FCN_PN_PRE=$v"; timeout 120 python tools/pn_bwd_stamps.py 2>&1 | tail -8 | tee $O/r05_b_pn_bwd_stamps_$v.txt
  echo "== isolated kernels FCN_PN_PRE=$v"; cd /tmp; rm -rf /tmp/prof_m$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m$v -o b -- python $R/tools/pn_micro.py 10 > /dev/null 2>&1
  cd $R; for f in $(find /tmp/prof_m$v -name "*kernel_stats*.csv"); do cp $f $O/r05_b_micro_${v}_kernel_stats.csv; done
done
python tools/kernel_compare.py $O/r05_b_micro_1_kernel_stats.csv $O/r05_b_micro_0_kernel_stats.csv | head -40
