for v in 0 1 0 1; do
  echo "FCN_NO_IOU=$v: $(FCN_NO_IOU=$v timeout 90 python bench.py --no-cpu-baseline --no-roofline --no-configs --min-time 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'])")"
done
