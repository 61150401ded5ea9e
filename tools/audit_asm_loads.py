#!/usr/bin/env python
"""Audit of the spconv kernels' inline-asm buffer loads (cdna_hip_programming.md 5.7): between an
asm `buffer_load_dwordx4 v[a:b]` and the counted `s_waitcnt vmcnt(N)` that retires it, no other
instruction may read or write v[a:b] (the compiler does not know the load is still in flight).
Linear scan of the .s in program order per kernel (FIFO retirement, loop back-edges ignored).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -c lidiff_amd/csrc/spconv.hip -save-temps -o /tmp/x.o
    python tools/audit_asm_loads.py spconv-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys


def regs_of(text):
    r = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        r.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        r.add(int(a))
    return r


def audit(name, lines):
    fifo, bad = [], 0
    for i, l in enumerate(lines):
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        m = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", t)
        if m:
            fifo.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            fifo = fifo[len(fifo) - n:] if n else []
            continue
        inflight = set().union(*fifo) if fifo else set()
        hit = regs_of(t.split(None, 1)[1] if " " in t else "") & inflight
        if hit:
            bad += 1
            print(f"  {name}: line {i}: `{t}` touches in-flight v{sorted(hit)[:4]}")
    print(f"{name}: {'CLEAN' if bad == 0 else str(bad) + ' VIOLATIONS'}")
    return bad


def main():
    s = open(sys.argv[1]).read()
    total = 0
    for m in re.finditer(r"^(_ZN6lidiff\w+):.*?\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        if "buffer_load_dwordx4" in m.group(2):
            total += audit(m.group(1)[24:60], m.group(2).split("\n"))
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
