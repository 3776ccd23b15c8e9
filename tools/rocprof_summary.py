#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite .db, the default output of rocprofv3 in ROCm 7.2):
per-kernel call count / avg / min / max duration, plus PMC counter sums per kernel if present.

    python tools/rocprof_summary.py <dir-or-db> [--filter cf] [--md]
"""
import glob
import os
import sqlite3
import sys


def find_db(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    if not hits:
        raise SystemExit(f"no .db under {path}")
    return hits[-1]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != flt]
    db = find_db(args[0])
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e3, "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[5] for r in rows) or 1.0
    print(f"# rocprofv3 kernel-trace summary ({os.path.basename(db)})\n")
    print("| kernel | calls | avg us | min us | max us | % of GPU time | VGPR | SGPR | LDS B | grid.x | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if flt and flt not in r[0]:
            continue
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2]:.2f} | {r[3]:.2f} | {r[4]:.2f} | {100 * r[5] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
    try:
        pm = list(c.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name"))
    except sqlite3.Error as e:
        pm = []
        print(f"\n(no PMC data: {e})")
    if pm:
        print("\n| kernel | counter | dispatches | avg per dispatch | min | max |")
        print("|---|---|---|---|---|---|")
        for r in pm:
            if flt and flt not in r[0]:
                continue
            name = r[0] if len(r[0]) < 70 else r[0][:67] + "..."
            print(f"| `{name}` | {r[1]} | {r[2]} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} |")


if __name__ == "__main__":
    main()
