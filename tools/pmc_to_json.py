#!/usr/bin/env python3
"""rocprofv3 --pmc passes (rocpd databases) of the 4K DIBR-only run -> profiles/pmc_latest.json, the per-launch figures bench.py
attaches to its roofline objects: corrected HBM/fabric bytes (2 x FETCH_SIZE + WRITE_SIZE, KB -> B; the x2 read correction is the
gfx950 one calibrated on k_stream_copy, MI355X_MICROARCH.md "HBM") and VALU lane-instructions (SQ_INSTS_VALU x 64).

    python tools/pmc_to_json.py r02 fetch.db write.db sq.db > profiles/pmc_latest.json"""
import json
import sqlite3
import sys
from collections import defaultdict

tag, paths = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for p in paths:
    for name, counter, value in sqlite3.connect(p).execute("select kernel_name, counter_name, value from counters_collection"):
        a = acc[name][counter]
        a[0] += value; a[1] += 1
avg = lambda n, c: (acc[n][c][0] / acc[n][c][1]) if acc[n].get(c) and acc[n][c][1] else None
out = {}


def entry(names):
    """per-launch figures summed over the launches that make up one stage (W1 = k_e2w + k_warp_fused since round 4)"""
    names = [n for n in names if n]
    if not names:
        return None
    tot = lambda c: (sum(avg(n, c) for n in names) if all(avg(n, c) is not None for n in names) else None)
    f, w, v = tot("FETCH_SIZE"), tot("WRITE_SIZE"), tot("SQ_INSTS_VALU")
    return {"kernel": " + ".join(n[:48] for n in names), "fetch_size_kb_raw": f, "write_size_kb": w,
            "corrected_bytes_per_launch": int(2 * f * 1024 + w * 1024) if f is not None and w is not None else None,
            "valu_wave_instr_per_launch": v, "valu_lane_instr_per_launch": v * 64 if v is not None else None,
            "salu_wave_instr_per_launch": tot("SQ_INSTS_SALU"),
            "lds_wave_instr_per_launch": tot("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": tot("SQ_LDS_BANK_CONFLICT"),
            "source": f"profiles/{tag}_pmc_4k_dibr.md"}


first = lambda *pre: next((k for p_ in pre for k in acc if k.startswith(p_)), None)
e = entry([first("k_e2w"), first("void k_warp_fused<true, true", "void k_warp_fused<")])
if e:
    out["k_warp_fused"] = e
e = entry([first("void k_finish_fused<true, 26", "void k_finish_fused<true", "void k_finish_fused<false", "k_finish_fused")])
if e:
    out["k_finish_fused"] = e
out["kernels"] = {n[:60]: {c: avg(n, c) for c in sorted(acc[n])} for n in sorted(acc) if n.startswith(("k_", "void k_"))}
import os
print(json.dumps({"4k-dibr": out, "commit": os.environ.get("VD3D_COMMIT", "unknown"), "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in three separate passes over "
                  "`bench.py --workload 4k-dibr`; read side x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request; calibrated on k_stream_copy)"}, indent=1))
