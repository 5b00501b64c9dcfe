#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_final.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_final.txt | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
TAG=${TAG:-r02_h} bash tools/gpu_final.sh
