#!/bin/bash
# Round-end measurement on one MI355X box, everything into gpurun_out/final_*: PMC passes first (bench.py then finds a
# traffic record that matches the kernel sources), the full bench line (roofline table + CPU baseline), the other configs and
# precisions, rocprofv3 kernel stats of the same command, phase stamps.  TAG names the files (default r02_final).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${TAG:-r02_final}
CFGS="${TRAFFIC_CFGS:-car people refine sunrgbd}" bash tools/gpu_traffic.sh > $O/${T}_traffic.log 2>&1; tail -3 $O/${T}_traffic.log | cut -c1-300
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json     # (on the box only: lets the bench below print traffic; merged back via gpurun_out)
cp $O/pmc_traffic.json $O/${T}_pmc_traffic.json; cp $O/pmc_sq_summary.txt $O/${T}_pmc_sq_summary.txt
cd $R
echo "== bench car (full, default invocation)"; timeout 600 python bench.py > $O/${T}_bench_car.json 2> $O/${T}_bench_car.err; echo "rc=$?"; tail -1 $O/${T}_bench_car.json | cut -c1-300
for c in people refine sunrgbd; do
  echo "== bench $c"; timeout 600 python bench.py --cfg $c --steps 100 --warmup 20 --no-cpu-baseline --no-configs > $O/${T}_bench_$c.json 2> $O/${T}_bench_$c.err; echo "rc=$?"; tail -1 $O/${T}_bench_$c.json | cut -c1-200
done
for p in f32 bf16 bf16ops; do
  echo "== bench car $p"; timeout 600 python bench.py --precision $p --steps 200 --warmup 30 --no-cpu-baseline --no-configs > $O/${T}_bench_car_$p.json 2> $O/${T}_bench_car_$p.err; echo "rc=$?"; tail -1 $O/${T}_bench_car_$p.json | cut -c1-200
done
echo "== phase stamps"; timeout 300 python tools/phase_stamps.py > $O/${T}_phase_stamps.txt 2>&1; tail -9 $O/${T}_phase_stamps.txt
for c in car sunrgbd; do
  echo "== rocprof $c"; cd /tmp; rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --cfg $c --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/${T}_prof_bench_$c.txt 2> $O/${T}_prof_$c.err; echo "rc=$?"
  cd $R; for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $O/${T}_kernel_stats_$c.csv; done; for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do (head -1 $f; tail -1500 $f) > $O/kernel_trace_final_$c.csv; done
done
python tools/trace_summary.py $O/kernel_trace_final_car.csv 60 > $O/${T}_trace_summary_car.txt 2>&1; grep -i "copyBuffer\|fillBuffer\|step window" $O/${T}_trace_summary_car.txt
head -8 $O/${T}_kernel_stats_car.csv | cut -c1-170
