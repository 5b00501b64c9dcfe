#!/bin/bash
# full default bench line (what the driver runs) + timing of the whole command
mkdir -p gpurun_out; O=gpurun_out; export TMPDIR=/tmp
( time timeout 600 python bench.py > $O/r05_p_bench_full.json 2> $O/r05_p_bench_full.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_p_bench_full.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('fwd_bwd_excl_optimizer'))
print(json.dumps(d['roofline'])[:400])
for k in d['roofline'].get('kernels', []): print(k.get('entry'), k.get('ms_per_step'), k.get('frac'))
for c in d.get('configs', []): print(c.get('cfg'), c.get('precision', c.get('mode')), c.get('value'), c.get('ms_per_step'), c.get('roofline', {}).get('frac'), c.get('error'))
print(d.get('cpu_baseline')); print(d.get('hbm_roofline'))
PY
tail -3 $O/r05_p_bench_full.err
