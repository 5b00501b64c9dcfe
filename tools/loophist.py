import sys,re,collections
fn, label = sys.argv[1], sys.argv[2]
lines=open(fn).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(label+':')][0]
end=next(i for i in range(start,len(lines)) if 's_endpgm' in lines[i])
body=lines[start:end]
# find loops: label positions and backward branches
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
loops=[]
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        loops.append((labels[m.group(1)],i))
print('kernel lines',len(body),'loops',loops)
for a,b in loops:
    if b-a<60: continue
    c=collections.Counter()
    for l in body[a:b+1]:
        t=l.strip().split()
        if not t or t[0].startswith(('.',';')): continue
        op=t[0]
        if op.startswith('v_mfma'): k='MFMA'
        elif op.startswith('v_'): k='VALU'
        elif op.startswith('ds_read') or op.startswith('ds_load'): k='ds_read:'+op
        elif op.startswith('ds_'): k='ds_write:'+op
        elif op.startswith('global_load') or op.startswith('buffer_load'): k='gload:'+op
        elif op.startswith('global_store'): k='gstore'
        elif op.startswith('s_waitcnt'): k='waitcnt'
        elif op.startswith('s_barrier'): k='barrier'
        elif op.startswith('s_'): k='SALU'
        else: k=op
        c[k]+=1
    print('loop',a,b,dict(c))
