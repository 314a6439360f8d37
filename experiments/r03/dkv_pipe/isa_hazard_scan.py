import re,sys
lines=[l.rstrip() for l in open(sys.argv[1])]
kern=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith('_Z') and kern in l and (l.endswith(':') or '; @' in l)][0]
end=next(i for i in range(start,len(lines)) if '.end_amdhsa_kernel' in lines[i])
ins=[]
for l in lines[start:end]:
    m=re.match(r'^\s+([vs]_\w+|ds_\w+|buffer_\w+)\s*(.*)$', l)
    if m: ins.append((m.group(1), m.group(2)))
def regs(tok):
    out=set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1): out |= {(m.group(1),k) for k in range(int(m.group(2)),int(m.group(3))+1)}
        else: out.add((m.group(4),int(m.group(5))))
    return out
bad={}
for i,(op,args) in enumerate(ins):
    if not op.startswith('v_mfma'): continue
    a=[x.strip() for x in args.split(',')]
    src=set()
    for t in a[1:]: src|=regs(t)
    ws=0
    for j in range(i-1,max(i-6,-1),-1):
        o,ar=ins[j]
        if o=='s_nop':
            ws+=int(ar.strip())+1; continue
        if ws>=2: break
        if o.startswith('v_') and not o.startswith('v_mfma'):
            dst=regs(ar.split(',')[0])
            if dst & src:
                bad.setdefault((o,ws),0); bad[(o,ws)]+=1
        ws+=1
print(bad)
# MFMA dst read by a following instruction within W wait states
W=12
bad2={}
for i,(op,args) in enumerate(ins):
    if not op.startswith('v_mfma'): continue
    dst=regs(args.split(',')[0])
    ws=0
    for j in range(i+1,min(i+40,len(ins))):
        o,ar=ins[j]
        if o=='s_nop': ws+=int(ar.strip())+1; continue
        if ws>=W: break
        parts=[x for x in ar.split(',')]
        srcs=set()
        for t in (parts[1:] if not o.startswith('ds_') and not o.startswith('buffer') else parts): srcs|=regs(t)
        if o.startswith('v_mfma'):
            # dependent chain allowed (C operand) but A/B operand reads are hazards
            ab=set()
            for t in parts[1:3]: ab|=regs(t)
            if ab & dst: bad2.setdefault(('mfma A/B reads mfma dst',ws),0); bad2[('mfma A/B reads mfma dst',ws)]+=1
            ws+=8; continue
        if srcs & dst:
            bad2.setdefault((o,ws),0); bad2[(o,ws)]+=1
        d2=regs(parts[0]) if parts else set()
        if d2 & dst and not o.startswith('v_mfma'):
            bad2.setdefault(('WAW '+o,ws),0); bad2[('WAW '+o,ws)]+=1
        ws+=1
print("mfma dst hazards:", bad2)
print("---- contexts")
for i,(op,args) in enumerate(ins):
    if not op.startswith('v_mfma'): continue
    dst=regs(args.split(',')[0])
    ws=0
    for j in range(i+1,min(i+30,len(ins))):
        o,ar=ins[j]
        if o=='s_nop': ws+=int(ar.strip())+1; continue
        if ws>=12: break
        if o.startswith('v_mfma'): ws+=8; continue
        d2=regs(ar.split(',')[0])
        if d2 & dst:
            print("MFMA", args, "| later", o, ar, "ws", ws, "idx", i)
        ws+=1
