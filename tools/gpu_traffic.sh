#!/bin/bash
# HBM traffic + matrix-pipe counters of one replayed step: SEPARATE rocprofv3 --pmc passes over the bench command itself
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters never combined with sys/hip/hsa tracing), summarised into
# gpurun_out/pmc_traffic.json by tools/pmc_summarize.py (copy to profiles/ when the kernel sources are final).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
# CFGS="car people refine sunrgbd": one pair of passes per configuration, all into the one cfg-keyed record
rm -f $O/pmc_traffic.json
for cfg in ${CFGS:-car}; do
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcs_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$c -o p -- python $R/bench.py --cfg $cfg --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmcs_${c}_$cfg.log 2>&1; echo "$cfg step $c rc=$?"
    f=$(find /tmp/pmcs_$c -name "*counter_collection.csv" | head -1); cp "$f" $O/pmcs_${c}_$cfg.csv 2>/dev/null
    wc -l $O/pmcs_${c}_$cfg.csv
  done
  cd $R; python tools/pmc_summarize.py $O/pmcs_FETCH_SIZE_$cfg.csv $O/pmcs_WRITE_SIZE_$cfg.csv $O/pmc_traffic.json $cfg
  [ $cfg != car ] && rm -f $O/pmcs_FETCH_SIZE_$cfg.csv $O/pmcs_WRITE_SIZE_$cfg.csv      # (the car passes stay as the raw record)
done
cd /tmp; rm -rf /tmp/pmcs_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcs_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmcs_sq.log 2>&1; echo "step SQ rc=$?"
f=$(find /tmp/pmcs_sq -name "*counter_collection.csv" | head -1); cp "$f" $O/pmcs_sq.csv 2>/dev/null
cd $R; python tools/pmc_sq_summary.py $O/pmcs_sq.csv > $O/pmc_sq_summary.txt 2>&1; tail -30 $O/pmc_sq_summary.txt
