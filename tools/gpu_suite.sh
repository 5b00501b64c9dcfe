#!/bin/bash
# the whole GPU suite + smoke on one box (TAG names the records)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; T=${TAG:-suite}
timeout 1500 python -m pytest tests -q -m gpu > $O/${T}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -4 $O/${T}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/${T}_smoke.txt
