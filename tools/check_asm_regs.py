import re, sys
lines=open(sys.argv[1]).read().split('\n')
def regs(tok):
    mm=re.match(r'v\[?(\d+)(?::(\d+))?\]?$', tok.strip())
    if not mm: return set()
    a=int(mm.group(1)); b=int(mm.group(2)) if mm.group(2) else a
    return set(range(a,b+1))
pending={}   # reg -> line of asm read
inasm=False; bad=[]
for n,l in enumerate(lines):
    s=l.split(';')[0].strip()
    if 'ASMSTART' in l: inasm=True; continue
    if 'ASMEND' in l: inasm=False; continue
    if not s: continue
    if inasm:
        m=re.match(r'ds_read_b(128|64|32) (\S+),', s)
        if m:
            for r in regs(m.group(2).rstrip(',')): pending[r]=n+1
        if s.startswith('s_waitcnt'):
            if 'lgkmcnt(0)' in s: pending={}
            else:
                # counted wait: everything but the newest k requests retired -> keep only newest k destination groups
                mk=re.search(r'lgkmcnt\((\d+)\)', s)
                if not mk: continue
                k=int(mk.group(1))
                keep=sorted(set(pending.values()))[-k:] if k else []
                pending={r:v for r,v in pending.items() if v in keep}
        continue
    if s.startswith('s_waitcnt') and 'lgkmcnt(0)' in s: pending={}; continue
    if s.startswith('v_mfma'): 
        parts=s.split(None,1)[1].split(',')
        for o in parts[1:3]:
            for r in regs(o):
                if r in pending: bad.append((n+1,s,'MFMA reads in-flight v%d (asm read at line %d)'%(r,pending[r])))
        continue
    parts=s.split(None,1)
    if len(parts)<2 or parts[0].startswith(('s_','.','ds_read')): continue
    ops=[o.strip() for o in parts[1].split(',')]
    srcs=ops if parts[0].startswith(('ds_write','buffer_store','global_store','buffer_load','ds_add')) else ops[1:]
    for o in srcs:
        for r in regs(o):
            if r in pending: bad.append((n+1,s,'reads in-flight v%d (asm read at line %d)'%(r,pending[r])))
print(len(bad),'suspicious reads of in-flight asm-loaded registers (linear scan, ignores control flow)')
for b in bad[:50]: print(b)
