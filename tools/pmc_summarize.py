"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes over `bench.py` (tools/gpu_traffic.sh): HBM bytes per step and
per launch, by kernel and by C-ABI entry point.

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
passes, both are in KB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (x2 correction);
WRITE_SIZE is taken as reported.  One step = the kernels between two consecutive loss-tail launches of the replayed graph
(the last complete window of the run, i.e. a timed step, not the warm-up).

The record is keyed by CONFIGURATION (car | people | refine | sunrgbd): each call adds / replaces one configuration's entry of
out.json (entries measured on other kernel sources are dropped), so bench.py --cfg X reports X's own traffic or null.

usage: python tools/pmc_summarize.py gpurun_out/pmcs_FETCH_SIZE.csv gpurun_out/pmcs_WRITE_SIZE.csv out.json [cfg = car]
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel name prefix -> C-ABI entry point whose call launches it (include/fcn_hip.h)
ENTRY = (("gc_hits_kernel", "fcn_pn_group_compact2"), ("gc_entries_kernel", "fcn_pn_group_compact2"),
         ("gc_fold_kernel", "fcn_pn_group_compact2"), ("pn_mid_kernel", "fcn_pn_backward2"),
         ("fwd_gemm_kernel", "fcn_pn_forward"), ("pool_nlc_kernel", "fcn_pn_forward"), ("pool_kernel", "fcn_pn_forward"),
         ("bn_finalize", "fcn_pn_forward"), ("poolbwd", "fcn_pn_backward2"), ("dgrad_kernel", "fcn_pn_backward2"),
         ("wgrad_kernel", "fcn_pn_backward2"), ("wgrad_reduce", "fcn_pn_backward2"), ("l1_finalize", "fcn_pn_backward2"),
         ("bnbwd", "fcn_pn_backward2"), ("cg_pack_kernel", "fcn_convnet_pack"), ("cgk_fwd", "fcn_convnet_forward2"),
         ("cg_bwd", "fcn_convnet_backward"), ("loss_tail_kernel", "fcn_det_loss_tail_rows2"),
         ("iou_metric_kernel", "fcn_det_iou_metrics"), ("adam_kernel", "fcn_adam_step_f32"))


def entry_of(name):
    for k, e in ENTRY:
        if k in name:
            return e
    return "other"


def step_window(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("loss_tail_kernel") or
             "loss_tail_kernel" in r["Kernel_Name"]]
    if len(marks) < 3:
        raise SystemExit("%s: fewer than 3 loss-tail launches in the trace" % path)
    a, b = marks[-2], marks[-1]
    return rows[a:b]


def main():
    fpath, wpath, out = sys.argv[1:4]
    cfg_name = sys.argv[4] if len(sys.argv) > 4 else "car"
    from bench import source_hash
    kern = collections.OrderedDict()
    for path, counter, corr in ((fpath, "FETCH_SIZE", 2.0), (wpath, "WRITE_SIZE", 1.0)):
        for r in step_window(path, counter):
            name = r["Kernel_Name"].split("(")[0]
            d = kern.setdefault(name, {"launches": 0, "FETCH_SIZE_kb_raw": 0.0, "WRITE_SIZE_kb_raw": 0.0})
            if counter == "FETCH_SIZE":
                d["launches"] += 1
            d[counter + "_kb_raw"] += float(r["Counter_Value"])
    entries = collections.OrderedDict()
    tot = 0.0
    for name, d in kern.items():
        d["bytes_per_step"] = int(1024 * (2.0 * d["FETCH_SIZE_kb_raw"] + d["WRITE_SIZE_kb_raw"]))
        tot += d["bytes_per_step"]
        e = entries.setdefault(entry_of(name), {"launches_per_step": 0, "bytes_per_step": 0})
        e["launches_per_step"] += d["launches"]
        e["bytes_per_step"] += d["bytes_per_step"]
    for e in entries.values():
        e["bytes_per_launch"] = int(e["bytes_per_step"] / max(e["launches_per_step"], 1))
    res = {"source_hash": source_hash(), "configs": {}}
    try:
        old = json.load(open(out))
        if old.get("source_hash") == res["source_hash"] and isinstance(old.get("configs"), dict):
            res["configs"] = old["configs"]
    except Exception:  # noqa
        pass
    res["command"] = ("rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --cfg <cfg> --steps 4 --warmup 2 "
                      "--no-cpu-baseline --no-roofline   (two separate passes per configuration, tools/gpu_traffic.sh)")
    res["note"] = ("bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): both counters are KB; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads), WRITE_SIZE as reported; one step = the "
                   "kernels between the last two loss-tail launches of the replayed hipGraph (B = 32 frustums); Infinity-Cache "
                   "hits are counted by these counters")
    res["configs"][cfg_name] = {
        "precision": os.environ.get("FCN_PRECISION", "split"),
        "step": {"bytes_per_step": int(tot), "fetch_size_kb_raw": round(sum(d["FETCH_SIZE_kb_raw"] for d in kern.values()), 1),
                 "write_size_kb_raw": round(sum(d["WRITE_SIZE_kb_raw"] for d in kern.values()), 1)},
        "entries": entries,
        "kernels": kern,
    }
    json.dump(res, open(out, "w"), indent=1)
    print("%s: step bytes %.1f MB over %d kernels; by entry: %s" % (
        cfg_name, tot / 1e6, sum(d["launches"] for d in kern.values()),
        ", ".join("%s %.1f MB" % (k, v["bytes_per_step"] / 1e6) for k, v in entries.items())))


if __name__ == "__main__":
    main()
