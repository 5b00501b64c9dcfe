#!/bin/bash
# HBM traffic of the dominant kernel: two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/traffic_driver.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1; echo "$c rc=$?"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv 2>/dev/null
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log; head -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv | cut -c1-400
done
# whole step: the same two passes over the bench command itself (graph replay)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmcs_$c.log 2>&1; echo "step $c rc=$?"
  f=$(find /tmp/pmcs_$c -name "*counter_collection.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/pmcs_$c.csv 2>/dev/null
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmcs_$c.log | cut -c1-200; wc -l $GRAFT_REPO_ROOT/gpurun_out/pmcs_$c.csv
done
