"""Per-kernel table of every counter of one rocprofv3 --pmc pass (tools/pmc_pass.sh), over ONE step of the replayed graph (the
kernels between the last two loss-tail launches): launches, then each counter summed over the kernel's launches of the step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    d = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = [by[k] for k in sorted(by)]
marks = [i for i, d in enumerate(disp) if "loss_tail_kernel" in d["name"]]
a, b = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(disp))
names = sorted({k for d in disp[a:b] for k in d if k != "name"})
agg = collections.OrderedDict()
for d in disp[a:b]:
    c = agg.setdefault(d["name"][:44], collections.Counter())
    c["n"] += 1
    for k in names:
        c[k] += d.get(k, 0.0)
print("%-44s %3s " % ("kernel (one step)", "n") + " ".join("%14s" % k[-14:] for k in names))
key = names[0] if names else "n"
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0)):
    print("%-44s %3d " % (n, c["n"]) + " ".join("%14.4g" % c[k] for k in names))
print("(columns: " + ", ".join(names) + ")")
