#!/bin/bash
# instruction-fetch counters per kernel over a short bench run (PMC pass of its own: --kernel-trace + --pmc only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmci; timeout 600 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU --output-format csv -d /tmp/pmci -o p -- python $R/bench.py --steps 6 --warmup 2 --min-time 0 --no-cpu-baseline --no-roofline --no-configs > $O/ifetch.log 2>&1; echo rc=$?
f=$(find /tmp/pmci -name "*counter_collection.csv" | head -1); ls -la $f
python - "$f" > $O/ifetch_summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
rows = []
for k, c in agg.items():
    if c.get("SQ_IFETCH", 0) <= 0: continue
    rows.append((c["SQ_WAVE_CYCLES"], k, c))
rows.sort(reverse=True)
print("%-70s %6s %9s %9s %8s %8s %8s" % ("kernel", "calls", "ifetch/wv", "lat(cyc)", "wait%", "valu/wv", "cyc/wv"))
for _, k, c in rows[:24]:
    w = max(c["SQ_WAVES"], 1.0)
    print("%-70s %6d %9.0f %9.1f %8.1f %8.0f %8.0f" % (k, n[k], c["SQ_IFETCH"] / w, c["SQ_IFETCH_LEVEL"] / max(c["SQ_IFETCH"], 1), 100 * c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_INSTS_VALU"] / w, 4 * c["SQ_WAVE_CYCLES"] / w))
PY
cat $O/ifetch_summary.txt | cut -c1-150
