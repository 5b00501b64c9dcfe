#!/bin/bash
# One GPU session of the kernel work: [FULL=1: the whole -m gpu suite | else the model / PointNet / train-state tests] with the
# product library, then (VARIANT set) the one-box A/B of tools/gpu_ab_variant.sh against libfcn_hip_<VARIANT>.so.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; T=${TAG:-s}
if [ -n "$FULL" ]; then SEL="tests"; else SEL=${TESTS:-"tests/test_gpu_model.py tests/test_gpu_pointnet.py tests/test_gpu_train_state.py tests/test_gpu_properties.py tests/test_gpu_group_compact.py"}; fi
echo "== pytest $SEL"; timeout 1500 python -m pytest $SEL -m gpu -q -rP --durations=8 --timeout 600 > $O/pytest_$T.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/pytest_$T.txt | tail -3 | cut -c1-200; grep -E "^FAILED|^ERROR" $O/pytest_$T.txt | head -10
if [ -n "$VARIANT" ]; then SKIP_TESTS=1 bash tools/gpu_ab_variant.sh; fi
