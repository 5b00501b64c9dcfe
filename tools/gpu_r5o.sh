#!/bin/bash
# Round 5, session o: LDS padding of the 16-deep images (FCN backward) -- bench A/B against the previous library + SQ counters
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused_convnet or train_eval_parity" 2>&1 | tail -2
for i in 1 2 3; do
  for lib in libfcn_hip.so libfcn_hip_prev.so; do
    FCN_LIB_NAME=$lib timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/r05_o_${lib}_$i.json 2> $O/r05_o_err.txt
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/r05_o_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $O/r05_o_bench.txt
  done
done
cd /tmp; rm -rf /tmp/pmcs_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcs_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmcs_sq.log 2>&1; echo "step SQ rc=$?"
f=$(find /tmp/pmcs_sq -name "*counter_collection.csv" | head -1); cp "$f" $O/pmcs_sq.csv 2>/dev/null
cd $R; python tools/pmc_sq_summary.py $O/pmcs_sq.csv > $O/r05_o_pmc_sq_summary.txt 2>&1; head -14 $O/r05_o_pmc_sq_summary.txt
