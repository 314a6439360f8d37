import re,collections,sys
txt=open(sys.argv[1]).read()
kern=sys.argv[2]
lines=txt.splitlines()
start=[i for i,l in enumerate(lines) if l.startswith('_Z') and kern in l and l.rstrip().endswith(':') or (l.startswith('_Z') and kern in l and '; @' in l)]
start=start[0]
end=next(i for i in range(start,len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
blocks=[]; cur=[]; name='entry'
for l in lines[start+1:end]:
    m=re.match(r'^(\.LBB\w+):', l)
    if m: blocks.append((name,cur)); cur=[]; name=m.group(1)
    else: cur.append(l)
blocks.append((name,cur))
for name,b in blocks:
    ops=collections.Counter()
    for l in b:
        m=re.match(r'\s+(v_\w+|s_waitcnt|ds_\w+|buffer_\w+|s_barrier|s_cbranch\w+|s_\w+)', l)
        if m: ops[m.group(1)]+=1
    nm=sum(v for k,v in ops.items() if 'mfma' in k)
    nv=sum(v for k,v in ops.items() if k.startswith('v_') and 'mfma' not in k)
    if nm or nv>20:
        print(f"{name:12s} insns {len(b):4d} mfma {nm:3d} valu {nv:4d} | "+', '.join(f"{k}:{v}" for k,v in ops.most_common(12)))
