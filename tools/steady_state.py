#!/usr/bin/env python3
"""Per-step kernel breakdown of the last N steady-state steps of a rocprofv3 kernel trace of bench.py
(step marker = the pass-2 hand-off kernel, one per step).  usage: steady_state.py <results.db> [n_steps] [top]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name,start,end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if r[0].startswith("void k_handoff<false>")]
if len(marks) <= n:
    marks = [i for i, r in enumerate(rows) if r[0].startswith("k_chain_ingest")][::16]
sel = rows[marks[-n - 1]:marks[-1]]
span = (sel[-1][2] - sel[0][1]) / 1e6
d = defaultdict(lambda: [0, 0.0])
for nm, s, e in sel:
    d[nm][0] += 1; d[nm][1] += (e - s) / 1e6
tot = sum(v[1] for v in d.values())
print(f"{n} steps: span {span/n:.3f} ms/step, summed kernel time {tot/n:.3f} ms/step, {len(sel)/n:.0f} launches/step")
for nm, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1]/n:8.3f} ms/step {v[0]/n:6.1f} calls  {nm[:120]}")
