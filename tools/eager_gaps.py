#!/usr/bin/env python
"""Inter-kernel gaps of a rocprofv3 --kernel-trace of `bench.py --only-eager` (rocpd .db): for every run of >= 64 consecutive launches
of one persistent kernel, the kernel's average duration and the gap between one launch's end and the next one's start -- eager
launches that are GPU-bound run back to back, host-bound ones leave the GPU idle between kernels.
    python tools/eager_gaps.py <dir-or-db>"""
import glob
import os
import sqlite3
import sys

import numpy as np

path = sys.argv[1]
db = path if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
runs, cur = [], []
for name, s, e in rows:
    if cur and (name != cur[0][0]):
        runs.append(cur)
        cur = []
    cur.append((name, s, e))
runs.append(cur)
print("| kernel | launches in the run | avg duration us | median gap to the next launch us | p90 gap us | period us |\n|---|---|---|---|---|---|")
for r in runs:
    if len(r) < 64 or "fused_decode" not in r[0][0]:
        continue
    s = np.array([x[1] for x in r], dtype=np.float64)
    e = np.array([x[2] for x in r], dtype=np.float64)
    gap = (s[1:] - e[:-1]) / 1e3
    name = r[0][0] if len(r[0][0]) < 60 else r[0][0][:57] + "..."
    print(f"| `{name}` | {len(r)} | {np.mean(e - s) / 1e3:.2f} | {np.median(gap):.2f} | {np.percentile(gap, 90):.2f} | {np.median(np.diff(s)) / 1e3:.2f} |")
