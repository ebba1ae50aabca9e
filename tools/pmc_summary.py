#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (rocpd databases) -> markdown table for profiles/.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db gpurun_out/pmc_sq/p_results.db

FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md "HBM"): the table shows the raw value and the x2-corrected bytes.
"""
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        db = sqlite3.connect(p)
        for name, counter, value, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            a = acc[name][counter]
            a[0] += value; a[1] += 1
            dur[name][0] += d; dur[name][1] += 1
    counters = sorted({c for k in acc.values() for c in k})
    names = sorted(acc, key=lambda n: -dur[n][0])
    print("| kernel | " + " | ".join(counters) + " |")
    print("|---|" + "---|" * len(counters))
    for n in names[:24]:
        if n.startswith("__amd") or "at::native" in n:
            continue
        row = []
        for c in counters:
            v = acc[n].get(c)
            row.append("-" if not v or not v[1] else f"{v[0] / v[1]:.4g}")
        print(f"| `{n[:70]}` | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
