#!/usr/bin/env python3
"""Per-kernel instruction statistics from a `hipcc -S --cuda-device-only` listing (ISA review helper).
usage: isa_stats.py listing.s kernel_name_substring [...]"""
import re
import sys

s = open(sys.argv[1]).read()
for k in sys.argv[2:]:
    for m in re.finditer(r'^(_Z[^\n:]*' + re.escape(k) + r'[^\n:]*):', s, re.M):
        body = s[m.end():]
        body = body[:body.index('s_endpgm')]
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
        valu = sum(l.startswith('v_') for l in lines)
        print("%s\n  instr %d  valu %d  branches %d  vmcnt-waits %d  lgkm-waits %d  scratch %d  ds %d  global %d" % (
            m.group(1)[:90], len(lines), valu, sum('s_cbranch' in l for l in lines), sum('vmcnt' in l for l in lines),
            sum('lgkmcnt' in l for l in lines), sum(l.startswith('scratch_') for l in lines),
            sum(l.startswith('ds_') for l in lines), sum(l.startswith('global_') for l in lines)))
        sc = [l for l in lines if l.startswith('scratch_')]
        if sc:
            print("  first scratch ops:", sc[:6])
