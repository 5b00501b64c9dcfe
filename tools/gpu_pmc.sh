#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|OccupancyPercent|LDSBankConflict)\b" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters.txt
grep -E "MFMA|LDS|WAIT|WAVE_CYCLES|BUSY_CYCLES|ACTIVE_INST|INSTS_VALU$|OCCUP" $GRAFT_REPO_ROOT/gpurun_out/counters.txt | head -60
rm -rf /tmp/pmc1; timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1; echo rc=$?
ls /tmp/pmc1 | head; cp /tmp/pmc1/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc1_counters.csv 2>/dev/null; tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc1.log
