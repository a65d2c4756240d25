import re,sys
lines=open(sys.argv[1]).read().split('\n')
# split functions
funcs=[];cur=None
for i,l in enumerate(lines):
    m=re.match(r'^(_Z\S+):',l)
    if m and 'gemm_sp_kernel' in m.group(1): cur=[m.group(1),i,None]; funcs.append(cur)
    if l.startswith('.Lfunc_end') and cur and cur[2] is None: cur[2]=i
for name,a,b in funcs:
    body=lines[a:b]
    # per basic block: count mfma and scratch
    blocks=[];bb=['entry',0,0,0,0]
    for l in body:
        m=re.match(r'^(\.LBB\S+):',l)
        if m: blocks.append(bb); bb=[m.group(1),0,0,0,0]
        if 'v_mfma' in l: bb[1]+=1
        if 'scratch_load' in l: bb[2]+=1
        if 'scratch_store' in l: bb[3]+=1
        if re.match(r'\s+[vsdgb]',l): bb[4]+=1
    blocks.append(bb)
    print(name[:90])
    for n,m,sl,ss,ins in blocks:
        if m or sl or ss: print(f"   {n:12s} ins {ins:5d} mfma {m:4d} scratch_load {sl:3d} scratch_store {ss:3d}")
