#!/bin/bash
# First GPU session after merging kernel changes: the whole -m gpu suite, smoke(), then the full measurement set under TAG
# (PMC passes -> profiles/pmc_traffic.json for the NEW kernel sources, bench lines of every configuration / precision, rocprofv3
# kernel stats, phase stamps).  ~9 GPU-minutes.   TAG=r03_a gpurun --timeout 1200 -- 'TAG=r03_a bash tools/gpu_round_start.sh'
# Afterwards: cp gpurun_out/${TAG}_* profiles/ ; cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q -rP --durations=15 --timeout 600 > gpurun_out/pytest_${TAG:-r03_a}.txt 2>&1; echo "rc=$?"; grep -E 'passed|failed|error' gpurun_out/pytest_${TAG:-r03_a}.txt | tail -3 | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
TAG=${TAG:-r03_a} bash tools/gpu_final.sh
