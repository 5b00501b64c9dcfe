#!/bin/bash
# Stream assignment of the captured step under the current FCN_* environment + a short bench: TAG names the outputs.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${TAG:-map}
rm -rf /tmp/dotw_$T; mkdir -p /tmp/dotw_$T; cd /tmp/dotw_$T
if [ -n "$OWN_CAPTURE" ]; then DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 200 python $R/tools/graph_dot.py capture > log.txt 2>&1
else DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 200 python $R/bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 0 --steps 4 --warmup 2 > log.txt 2>&1; fi   # bench.py's own captured step (two steps per graph, prefetch beside the FCN forward)
f=$(ls -S graph_*_dot_print_* 2>/dev/null | head -1)
cd $R; cp /tmp/dotw_$T/$f gpurun_out/${T}_graph.dot 2>/dev/null
python tools/graph_dot.py parse gpurun_out/${T}_graph.dot poolbwd dgrad wgrad l1_fin adam gc_ loss iou cg_pack fwd_gemm pool_nlc cgk_fwd cg_bwd > gpurun_out/${T}_map.txt 2>&1
for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$T', d['value'], d['ms_per_step'])"; done
