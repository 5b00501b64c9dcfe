#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; rm -rf /tmp/pmcs_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcs_sq -o p -- python $R/bench.py --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmcs_sq.log 2>&1; echo "step SQ rc=$?"
f=$(find /tmp/pmcs_sq -name "*counter_collection.csv" | head -1); cp "$f" $O/pmcs_sq.csv 2>/dev/null
cd $R; python tools/pmc_sq_summary.py $O/pmcs_sq.csv > $O/r05_o_pmc_sq_summary.txt 2>&1; head -14 $O/r05_o_pmc_sq_summary.txt
