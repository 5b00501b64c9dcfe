#!/bin/bash
# Round-2 GPU session J: SUN-RGBD fixture away from ReLU kinks, property tests for all four configs.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest gpu (model + properties)"; timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_properties.py -m gpu -q --timeout 600 -s > $O/pytest_j1.txt 2>&1; echo "rc=$?"; grep -E "sunrgbd|passed|failed|Error" $O/pytest_j1.txt | cut -c1-220 | tail -12
echo "== pytest gpu (rest)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_model.py --deselect tests/test_gpu_properties.py > $O/pytest_j2.txt 2>&1; echo "rc=$?"; tail -4 $O/pytest_j2.txt | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
