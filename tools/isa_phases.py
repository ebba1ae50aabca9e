#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing, cut at its s_barrier instructions (the phases of a tiled kernel are
separated by workgroup barriers).  Development aid for the per-phase VALU budgets in DESIGN.md.
usage: isa_phases.py file.s <kernel-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l][0]
end = [i for i, l in enumerate(lines) if 'NumVgprs' in l and i > start][0]
def new(): return dict(n=0, valu=0, pk=0, fma=0, trans=0, cvt=0, lds=0, vmem=0, salu=0, loops=0)
segs, cur = [], new()
for i in range(start + 1, end):
    l = lines[i]
    if re.match(r'^\.LBB\d+_\d+:', l):
        if 'Loop Header' in l: cur['loops'] += 1
        continue
    t = l.strip().split()[0] if l.strip() else ''
    if not t or t.startswith(';') or t.startswith('.'): continue
    if t == 's_barrier':
        segs.append(cur); cur = new(); continue
    cur['n'] += 1
    if t.startswith('v_'):
        cur['valu'] += 1
        if t.startswith('v_pk_'): cur['pk'] += 1
        if 'fma' in t or 'fmac' in t: cur['fma'] += 1
        if re.match(r'v_(rcp|sqrt|rsq|exp|log|sin|cos|div)', t): cur['trans'] += 1
        if t.startswith('v_cvt'): cur['cvt'] += 1
    elif t.startswith('ds_'): cur['lds'] += 1
    elif t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
    elif t.startswith('s_'): cur['salu'] += 1
segs.append(cur)
print(f"{'segment':8s} {'n':>6s} {'valu':>6s} {'pk':>5s} {'fma':>5s} {'trn':>4s} {'cvt':>4s} {'lds':>5s} {'vmem':>5s} {'salu':>6s} {'loops':>5s}")
for k, s in enumerate(segs):
    print(f"{k:8d} {s['n']:6d} {s['valu']:6d} {s['pk']:5d} {s['fma']:5d} {s['trans']:4d} {s['cvt']:4d} {s['lds']:5d} {s['vmem']:5d} {s['salu']:6d} {s['loops']:5d}")
tot = {k: sum(s[k] for s in segs) for k in segs[0]}
print('total  ', tot)
