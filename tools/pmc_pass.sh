#!/bin/bash
# One rocprofv3 --pmc pass (counters only + --kernel-trace, never combined with sys / hip / hsa tracing) over a short replayed bench run,
# per-kernel table of ONE step window -> gpurun_out/<TAG>_pmc_<name>.txt.   bash tools/pmc_pass.sh <TAG> <name> "<COUNTER ...>" [bench args]
# A counter the box does not know makes the whole pass fail: keep groups small, and `list` dumps the available names first.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; T=${1:?tag}; N=${2:?name}; C=${3:?counters}; shift 3
if [ "$N" = list ]; then
  cd /tmp; rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|SQC|TCP|TCC|TA|TD|GRBM|CPC|SPI)_[A-Za-z0-9_]+\b" | sort -u > $O/${T}_counters_avail.txt; wc -l $O/${T}_counters_avail.txt
  grep -E "^(TCP_(TCC|PENDING|TOTAL|TA_|GATE|READ|TCP)|TCC_(HIT|MISS|REQ|READ|EA0?_RDREQ|TAG_STALL|BUSY)|TA_(BUSY|TA_BUSY|ADDR_STALL|DATA_STALL|FLAT_READ|BUFFER_READ)|TD_(TD_BUSY|TC_STALL)|SQ_(WAIT_ANY|WAIT_INST_ANY|INSTS_VMEM|INST_CYCLES_VMEM|ACTIVE_INST_VMEM|INSTS_SMEM|WAVE_CYCLES|BUSY_CU))" $O/${T}_counters_avail.txt | tr '\n' ' '; echo
  exit 0
fi
cd /tmp; rm -rf /tmp/pmcp_$N
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcp_$N -o p -- python $R/bench.py --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs "$@" > $O/${T}_pmc_$N.log 2>&1; echo "pass $N rc=$?"
f=$(find /tmp/pmcp_$N -name "*counter_collection.csv" | head -1)
cd $R; python tools/pmc_table.py "$f" > $O/${T}_pmc_$N.txt 2>&1; head -${ROWS:-24} $O/${T}_pmc_$N.txt | cut -c1-${COLS:-230}
