#!/usr/bin/env python3
"""Static per-basic-block instruction census of one kernel in a hipcc -S listing (development aid for the VALU budgets in DESIGN.md).
usage: isa_blocks.py file.s <kernel-name-substring> [min_instructions]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l][0]
end = [i for i, l in enumerate(lines) if 'NumVgprs' in l and i > start][0]
stats = []
cur = None
def new(name, line):
    return dict(name=name, line=line, n=0, valu=0, pk=0, fma=0, trans=0, cvt=0, lds=0, vmem=0, salu=0, loop='')
cur = new('entry', start)
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        stats.append(cur); cur = new(m.group(1), i + 1)
        mm = re.search(r'Depth=(\d+)', m.group(2)); cur['loop'] = ('L' + mm.group(1)) if mm else ''
        continue
    t = l.strip().split()[0] if l.strip() else ''
    if not t or t.startswith(';') or t.startswith('.'):
        continue
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
stats.append(cur)
print(f"{'block':12s} {'line':>6s} {'loop':4s} {'n':>5s} {'valu':>5s} {'pk':>4s} {'fma':>5s} {'trn':>4s} {'cvt':>4s} {'lds':>4s} {'vmem':>4s} {'salu':>5s}")
for s in stats:
    if s['n'] >= mn:
        print(f"{s['name']:12s} {s['line']:6d} {s['loop']:4s} {s['n']:5d} {s['valu']:5d} {s['pk']:4d} {s['fma']:5d} {s['trans']:4d} {s['cvt']:4d} {s['lds']:4d} {s['vmem']:4d} {s['salu']:5d}")
tot = {k: sum(s[k] for s in stats) for k in ('n', 'valu', 'pk', 'fma', 'trans', 'cvt', 'lds', 'vmem', 'salu')}
print('total', tot)
