#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats`` output on ROCm 7.x) into the
per-kernel table that is committed under profiles/ (the .db itself stays in gpurun_out/ scratch).

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count),"
        " max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x*grid_y*grid_z), max(workgroup_x*workgroup_y*workgroup_z)"
        " from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path.split('/')[-1]}`\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / total:.1f} |"
              f" {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
