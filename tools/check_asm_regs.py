"""Checks over a gfx950 assembly listing (hipcc -S --cuda-device-only):
   python tools/check_asm_regs.py file.s            no instruction reads a register an inline-asm ds_read still has in flight
   python tools/check_asm_regs.py --no-fma file.s   no fused multiply-add anywhere (step.hip restates separately rounded torch
                                                    launches: a contracted a * b + c would change voxelisation at cell boundaries)"""
import re, sys
if "--no-fma" in sys.argv:
    path = [a for a in sys.argv[1:] if a != "--no-fma"][0]
    hits = [(n + 1, l.strip()) for n, l in enumerate(open(path).read().split('\n'))
            if re.match(r'\s*v_(pk_)?(fma|fmac|mad|mac)_(f|legacy_f)(16|32|64)', l)]
    print(len(hits), "fused multiply-add instructions")
    for n, l in hits[:20]:
        print("  line %d: %s" % (n, l))
    sys.exit(1 if hits else 0)
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
