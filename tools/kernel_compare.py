"""Per-kernel average durations (us) of several rocprofv3 kernel_stats csv files side by side (first file = reference)."""
import csv
import sys


def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
    return d


tabs = [load(p) for p in sys.argv[1:]]
names = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, (0, 0, 0))[2] for t in tabs))
print("%-44s %5s " % ("kernel", "calls") + " ".join("%9s" % p.split("/")[-1].replace("_kernel_stats.csv", "")[-9:] for p in sys.argv[1:]))
tot = [0.0] * len(tabs)
for k in names:
    if "at::" in k or "rocclr" in k:
        continue
    calls = max(t.get(k, (0, 0, 0))[0] for t in tabs)
    print("%-44s %5d " % (k[:44], calls) + " ".join("%9.1f" % t.get(k, (0, 0, 0))[1] for t in tabs))
    for i, t in enumerate(tabs):
        tot[i] += t.get(k, (0, 0, 0))[2]
print("%-44s %5s " % ("total kernel time (ms)", "") + " ".join("%9.2f" % (v / 1e3) for v in tot))
