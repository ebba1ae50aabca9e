#!/usr/bin/env python3
"""rocprofv3 passes (rocpd databases) of DIBR-only bench workloads -> profiles/pmc_latest.json, the per-launch figures bench.py attaches to its
roofline objects: corrected HBM / fabric bytes (2 x FETCH_SIZE + WRITE_SIZE, KB -> B; the x2 read correction is the gfx950 one calibrated on
k_stream_copy, MI355X_MICROARCH.md "HBM"), VALU lane-instructions (SQ_INSTS_VALU x 64) and -- round 5 -- the kernels' average durations from the
`--kernel-trace --stats` run of the same workload (`rocprof_avg_launch_us`: the figure the committed kernel-stats tables show; bench.py prints it next
to its own HIP-event figure and names both).

    python tools/pmc_to_json.py r05 4k-dibr=fetch.db,write.db,sq.db[,trace.db] 4k-dibr-gui=... > profiles/pmc_latest.json"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

tag, specs = sys.argv[1], sys.argv[2:]


def load(paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    durs = {}
    for p in paths:
        db = sqlite3.connect(p)
        tabs = {r[0] for r in db.execute("select name from sqlite_master")}
        if "counters_collection" in tabs:
            for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
                a = acc[name][counter]
                a[0] += value; a[1] += 1
        if "kernels" in tabs and not acc:   # a kernel-trace database without counters: durations only
            pass
        if "kernels" in tabs:
            try:
                for name, n, avg in db.execute("select name, count(*), avg(duration) from kernels group by name"):
                    durs.setdefault(name, (n, avg / 1e3))
            except sqlite3.Error:
                pass
    return acc, durs


def workload_entry(acc, durs, src):
    avg = lambda n, c: (acc[n][c][0] / acc[n][c][1]) if acc[n].get(c) and acc[n][c][1] else None
    first = lambda *pre: next((k for p_ in pre for k in acc if k.startswith(p_)), None)

    def entry(names):
        names = [n for n in names if n]
        if not names:
            return None
        tot = lambda c: (sum(avg(n, c) for n in names) if all(avg(n, c) is not None for n in names) else None)
        f, w, v = tot("FETCH_SIZE"), tot("WRITE_SIZE"), tot("SQ_INSTS_VALU")
        dur = [durs.get(n) for n in names]
        return {"kernel": " + ".join(n[:56] for n in names), "fetch_size_kb_raw": f, "write_size_kb": w,
                "corrected_bytes_per_launch": int(2 * f * 1024 + w * 1024) if f is not None and w is not None else None,
                "valu_wave_instr_per_launch": v, "valu_lane_instr_per_launch": v * 64 if v is not None else None,
                "salu_wave_instr_per_launch": tot("SQ_INSTS_SALU"),
                "lds_wave_instr_per_launch": tot("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": tot("SQ_LDS_BANK_CONFLICT"),
                "rocprof_avg_launch_us": (round(sum(d[1] for d in dur), 2) if all(dur) else None),
                "source": src}
    out = {}
    e = entry([first("k_e2w"), first("void k_warp_fused<true, true", "void k_warp_fused<")])
    if e:
        out["k_warp_fused"] = e
    e = entry([first("void k_finish_fused<true, 26", "void k_finish_fused<true", "void k_finish_fused<false", "k_finish_fused")])
    if e:
        out["k_finish_fused"] = e
    out["kernels"] = {n[:60]: dict({c: avg(n, c) for c in sorted(acc[n])}, **({"rocprof_avg_us": round(durs[n][1], 2)} if n in durs else {}))
                      for n in sorted(acc) if n.startswith(("k_", "void k_"))}
    return out


res = {}
for spec in specs:
    wl, paths = spec.split("=", 1)
    acc, durs = load([p for p in paths.split(",") if p])
    res[wl] = workload_entry(acc, durs, f"profiles/{tag}_pmc_{wl.replace('-', '_')}.md")
res["commit"] = os.environ.get("VD3D_COMMIT", "unknown")
res["method"] = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in three separate passes over `bench.py --workload <name>` (pixel overlap on), kernel durations from a "
                 "fourth `--kernel-trace --stats` pass of `--no-pixel-overlap`; read side x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request; calibrated on k_stream_copy)")
print(json.dumps(res, indent=1))
