#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_group_compact.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_q_grouping_row.txt
import torch, bench, json
dev = torch.device("cuda", 0)
for cfg in ("car", "people"):
    data = bench.make_data(cfg, 32, bench.CFGS[cfg][3], 1234, dev)
    print(cfg, json.dumps(bench.grouping_op_row(data, cfg)))
PY
timeout 120 python tools/qdp_time.py 2>&1 | tail -6 | tee -a gpurun_out/r05_q_grouping_row.txt
