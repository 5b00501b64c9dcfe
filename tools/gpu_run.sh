#!/bin/bash
# The ONE parameterised GPU-session runner (replaces the per-session scripts of earlier rounds).  Runs on the GPU box:
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <TAG> <step> [<step> ...]'
#
# Steps (run in the order given; every record goes to gpurun_out/<TAG>_*):
#   tests[=<pytest args>]                   -m gpu tests (default: the whole suite), summary on stdout
#   smoke                                   __graft_entry__.smoke()
#   bench=<name>[,VAR=V...][::<args>]       one bench.py line under the given environment -> <TAG>_bench_<name>.json
#   full=<name>[,VAR=V...][::<args>]        the same without the --no-* switches (roofline table, configs, CPU baseline)
#   ab=<rounds>:<specA>/<specB>[/...][::<args>]   alternating short bench lines of several environments on this one box;
#                                           spec = <name>[,VAR=V...]  (FCN_LIB_NAME=libfcn_hip_<x>.so selects a tuning build)
#   prof=<name>[,VAR=V...][::<args>]        rocprofv3 --kernel-trace --stats of a 20-step bench run -> <TAG>_kernel_stats_<name>.csv
#                                           (+ the trace summary of one step window)
#   phases=<name>[,VAR=V...]                tools/phase_stamps.py under the environment
#   py=<name>[,VAR=V...]::<script + args>   any tool of this directory, output -> <TAG>_<name>.txt
#   final                                   tools/gpu_final.sh (PMC passes, every config / precision, kernel stats) under TAG
# Example:  bash tools/gpu_run.sh r6a tests=tests/test_gpu_dist.py 'ab=3:n1/rccl1,FCN_BENCH_COMM=rccl1' prof=car
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=${1:?tag}; shift
QUICK="--no-cpu-baseline --no-roofline --no-configs --min-time 1.5"

line() { python - "$1" <<'EOF'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d.get("n_gpus"), d.get("rccl_ranks", ""))
except Exception as e:
    print("no line (%s)" % e)
EOF
}
# spec "name,VAR=V,VAR2=V2" -> NAME and the env words
parse() { NAME=${1%%,*}; ENVS=""; if [ "$1" != "$NAME" ]; then ENVS=$(echo "${1#*,}" | tr ',' ' '); fi; }

for step in "$@"; do
  kind=${step%%=*}; rest=""; [ "$step" != "$kind" ] && rest=${step#*=}
  args=""; if [[ "$rest" == *"::"* ]]; then args=${rest#*::}; rest=${rest%%::*}; fi
  case $kind in
    tests)
      sel=${rest:-tests}
      echo "== pytest $sel"; timeout 1700 python -m pytest $sel -m gpu -q -rP --durations=10 --timeout 900 > $O/${T}_pytest.txt 2>&1; echo "rc=$?"
      grep -E "passed|failed|error" $O/${T}_pytest.txt | tail -3 | cut -c1-200; grep -E "^FAILED|^ERROR" $O/${T}_pytest.txt | head -12 ;;
    smoke)
      echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
    bench|full)
      parse "$rest"; sw=$QUICK; [ $kind = full ] && sw=""
      echo "== $kind $NAME [$ENVS] $args"
      env $ENVS timeout 900 python bench.py $sw $args > $O/${T}_bench_$NAME.json 2> $O/${T}_bench_$NAME.err; echo "rc=$? $(line $O/${T}_bench_$NAME.json)" ;;
    ab)
      rounds=${rest%%:*}; specs=$(echo "${rest#*:}" | tr '/' ' ')
      for i in $(seq 1 $rounds); do
        for sp in $specs; do
          parse "$sp"
          env $ENVS timeout 120 python bench.py $QUICK $args > $O/${T}_ab_${NAME}_$i.json 2> $O/${T}_ab_${NAME}_$i.err
          echo "$NAME $i: $(line $O/${T}_ab_${NAME}_$i.json)"
        done
      done ;;
    prof)
      parse "$rest"; echo "== rocprof $NAME [$ENVS] $args"
      cd /tmp; rm -rf /tmp/prof_$NAME
      env $ENVS timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o b -- python $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-roofline --no-configs $args > $O/${T}_prof_$NAME.txt 2> $O/${T}_prof_$NAME.err; echo "rc=$?"
      cd $R
      for f in $(find /tmp/prof_$NAME -name "*kernel_stats*.csv"); do cp $f $O/${T}_kernel_stats_$NAME.csv; done
      for f in $(find /tmp/prof_$NAME -name "*kernel_trace*.csv"); do (head -1 $f; tail -1500 $f) > $O/${T}_kernel_trace_$NAME.csv; done
      python tools/trace_summary.py $O/${T}_kernel_trace_$NAME.csv 60 > $O/${T}_trace_summary_$NAME.txt 2>&1
      head -12 $O/${T}_kernel_stats_$NAME.csv | cut -c1-150 ;;
    phases)
      parse "$rest"; echo "== phases $NAME [$ENVS]"
      env $ENVS timeout 300 python tools/phase_stamps.py > $O/${T}_phases_$NAME.txt 2>&1; tail -9 $O/${T}_phases_$NAME.txt ;;
    py)
      parse "$rest"; echo "== py $NAME [$ENVS] $args"
      env $ENVS timeout 900 python $args > $O/${T}_$NAME.txt 2>&1; echo "rc=$?"; tail -${TAIL:-15} $O/${T}_$NAME.txt | cut -c1-220 ;;
    final)
      TAG=$T bash tools/gpu_final.sh ;;
    *) echo "unknown step '$step'"; exit 2 ;;
  esac
done
