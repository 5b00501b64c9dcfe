"""Prints the interesting fields of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms |", d["dtype"][:24], "|", d["config"]["launch"])
rl = d.get("roofline", {})
print("roofline", {k: v for k, v in rl.items() if k not in ("kernels", "traffic_note", "how")})
for r in rl.get("kernels", []):
    print("   %-62s ms %-8s frac %-7s %s" % (r["entry"][:62], r.get("ms_per_step"), r.get("frac"), r.get("achieved_tbps", r.get("achieved_tflops", ""))))
print("hbm", d.get("hbm_roofline"))
for c in d.get("configs", []):
    print("   cfg %-8s %-8s %-10s %-10s %-8s %s %s" % (c.get("cfg"), c.get("precision"), c.get("mode", "train"), c.get("value"), c.get("ms_per_step"), c.get("error", ""), c.get("workload", "")[:80]))
cb = d.get("cpu_baseline")
print("cpu", {k: v for k, v in cb.items() if k != "sample"} if cb else None)
if "comm" in d:
    print("comm", d["comm"])
