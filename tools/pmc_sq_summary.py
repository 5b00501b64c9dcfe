"""Per-kernel SQ counters of one replayed step (rocprofv3 --pmc pass of tools/gpu_traffic.sh): matrix-pipe busy share, wait
share, LDS bank-conflict share.  One step = the kernels between the last two loss-tail launches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_disp = collections.OrderedDict()
for r in rows:
    d = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
disp = [by_disp[k] for k in sorted(by_disp)]
marks = [i for i, d in enumerate(disp) if "loss_tail_kernel" in d["name"]]
a, b = marks[-2], marks[-1]
agg = collections.OrderedDict()
for d in disp[a:b]:
    n = d["name"][:56]
    c = agg.setdefault(n, collections.Counter())
    c["launches"] += 1
    for k, v in d.items():
        if k != "name":
            c[k] += v
print("%-56s %3s %10s %8s %8s %8s" % ("kernel", "n", "cu_busy", "mfma%", "wait%", "ldsconf%"))
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    busy = max(c.get("SQ_BUSY_CU_CYCLES", 0), 1)
    wave = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    lds = max(c.get("SQ_ACTIVE_INST_LDS", 0), 1)
    print("%-56s %3d %10.3g %8.1f %8.1f %8.1f" % (n, c["launches"], busy, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy,
                                                 100 * c.get("SQ_WAIT_INST_ANY", 0) / wave,
                                                 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds))
