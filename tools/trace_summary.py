"""Per-step summary of a rocprofv3 kernel trace (gpurun_out/kernel_trace.csv): kernel sums and phase spans."""
import csv, collections, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/kernel_trace.csv'
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'loss_tail_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
print("step window us", (int(rows[b]['Start_Timestamp']) - t0) / 1e3, "kernels", b - a)
agg = collections.OrderedDict()
for r in rows[a:b]:
    n = r['Kernel_Name'].split('(')[0][:50]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    c = agg.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-52s n=%3d sum=%8.1f" % (n, c, d))
def span(pred):
    xs = [(int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0) for r in rows[a:b] if pred(r['Kernel_Name'])]
    return (round(min(x[0] for x in xs) / 1e3, 1), round(max(x[1] for x in xs) / 1e3, 1)) if xs else None
print("fcn bwd", span(lambda n: 'cg_bwd_step' in n)); print("fcn fwd", span(lambda n: 'cgk_fwd' in n or 'cg_pack' in n))
print("pn bwd", span(lambda n: any(k in n for k in ('poolbwd', 'dgrad_kernel<', 'wgrad_kernel<', 'l1_finalize'))))
print("pn fwd", span(lambda n: any(k in n for k in ('qdp_kernel', 'fwd_gemm', 'pool_kernel'))))
print("adam", span(lambda n: 'adam' in n))
if len(sys.argv) > 3:
    for r in rows[a:b]:
        n = r['Kernel_Name']; s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
        if sys.argv[3] in n or sys.argv[3] == 'all':
            print("%8.1f %7.1f  %-58s g=%s,%s wg=%s" % (s, e - s, n[:58], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Workgroup_Size_X']))
