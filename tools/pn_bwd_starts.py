import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'loss_tail_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    n=r['Kernel_Name']
    if ('poolbwd' in n or 'adam' in n or 'l1_fin' in n or 'dgrad_kernel<2, 2, 1, 2' in n or ('cg_bwd' in n and s>400)):
        print("%7.1f %6.1f q=%s  %s g=%d,%s" % (s,e-s,r.get('Queue_Id'),n[:36],int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Grid_Size_Y']))
