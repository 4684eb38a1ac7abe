#!/usr/bin/env python3
"""Scan hipcc -S output for VALU reads of an MFMA result with fewer than 7 wait states, following
branches (found a missing s_nop pad on a fall-through path of conv1x1.hip with ROCm 7.2)."""
import sys,re
f=sys.argv[1]
raw=[l.rstrip() for l in open(f)]
lines=[]; labels={}
for l in raw:
    s=l.strip()
    if not s or s.startswith(';'): continue
    m=re.match(r'^(\.LBB\d+_\d+):',s)
    if m: labels[m.group(1)]=len(lines); continue
    if s.startswith('.') or s.endswith(':'): continue
    lines.append(s)
def regs(tok):
    tok=tok.split()[0] if tok else tok
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
bad=0
def scan(start,dst,waits,depth,origin):
    global bad
    j=start
    while j<len(lines) and waits<7 and depth<4:
        t=lines[j]
        if t.startswith('s_nop'): waits+=int(t.split()[1])+1; j+=1; continue
        if t.startswith('s_cbranch') or t.startswith('s_branch'):
            tgt=t.split()[1]
            if tgt in labels: scan(labels[tgt],dst,waits+1,depth+1,origin)
            if t.startswith('s_branch'): return
            waits+=1; j+=1; continue
        if t.startswith('v_mfma'):
            return
        if t.startswith('s_') or t.startswith('ds_') or t.startswith('global_') or t.startswith('buffer_') or t.startswith('scratch_'):
            # memory ops reading dst as data are also hazards but ignore
            waits+=1; j+=1; continue
        parts=t.split(None,1)
        ops=[x.strip() for x in parts[1].split(',')] if len(parts)>1 else []
        src=set()
        for o in ops[1:]: src|=regs(o)
        if src & dst:
            bad+=1; print(f.split('/')[-1],'MFMA@',origin,'-> read after',waits,'waits:',t[:70]); return
        if ops and (regs(ops[0]) & dst): dst=dst-regs(ops[0])
        if not dst: return
        waits+=1; j+=1
for i,l in enumerate(lines):
    if l.startswith('v_mfma'):
        ops=[t.strip() for t in l.split(None,1)[1].split(',')]
        scan(i+1,regs(ops[0]),0,0,i)
print(f.split('/')[-1],'bad',bad)
