#!/bin/bash
# A/B of a tuning build against the product library on ONE box: parity subset with the variant (TESTS, KEXPR), then alternating
# bench runs and the phase anatomy of both (SKIP_TESTS=1 / SKIP_PHASES=1 leave those parts out).  VARIANT names frustum_convnet_amd/libfcn_hip_<VARIANT>.so (tools/build_variant.py).
mkdir -p gpurun_out; O=gpurun_out; V=${VARIANT:-wg}; export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
echo "== parity subset with $V"
FCN_LIB_NAME=libfcn_hip_$V.so timeout 150 python -m pytest ${TESTS:-tests/test_gpu_model.py} -x -q ${KEXPR:+-k "$KEXPR"} > $O/ab_${V}_pytest.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_${V}_pytest.txt
fi
for i in 1 2; do
  for lib in prod $V; do
    if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
    timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 > $O/ab_${lib}_$i.json 2> $O/ab_${lib}_$i.err
    echo "$lib $i: $(python -c "import json,sys; d=json.loads(open('$O/ab_${lib}_$i.json').read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
if [ -z "$SKIP_PHASES" ]; then
for lib in prod $V; do
  if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
  timeout 90 python tools/phase_stamps.py > $O/ab_${lib}_phases.txt 2>&1; echo "-- phases $lib"; tail -8 $O/ab_${lib}_phases.txt
done
fi
if [ -n "$PROF" ]; then      # per-kernel averages of both libraries (rocprofv3 kernel trace over a short bench run)
for lib in prod $V; do
  if [ $lib = prod ]; then unset FCN_LIB_NAME; else export FCN_LIB_NAME=libfcn_hip_$lib.so; fi
  R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/prof_$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -o b -- python $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > /dev/null 2>&1
  cd $R; for f in $(find /tmp/prof_$lib -name "*kernel_stats*.csv"); do cp $f $O/ab_${lib}_kernel_stats.csv; done
  echo "-- kernel stats $lib"; head -14 $O/ab_${lib}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
fi
