#!/usr/bin/env python
"""Per-TCC-channel view of a rocprofv3 --pmc pass (rocpd .db): rocprofv3's summaries reduce a counter over its 128 instances
(16 TCC channels x 8 XCDs on gfx950); the raw rows are kept, in a fixed order per dispatch.  For every kernel and counter: mean per
dispatch of the sum, and how evenly the 128 channels share it (min / max over the channels' means, relative to the mean channel,
and the same per group of 16 consecutive rows = one XCD's channels if the rows are XCD-major).

    python tools/pmc_channels.py <db> [kernel-substring]"""
import collections
import sqlite3
import sys

import numpy as np

db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
names = dict(c.execute("select id, name from rocpd_info_pmc"))
kern = {}
for eid, kname in c.execute("select event_id, kernel_name from counters_collection"):
    kern[eid] = kname
rows = collections.defaultdict(list)      # (kernel, counter) -> list of per-dispatch vectors
cur = collections.defaultdict(list)
for eid, pid, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event order by id"):
    cur[(eid, pid)].append(val)
for (eid, pid), v in cur.items():
    k = kern.get(eid, "?")
    if flt in k:
        rows[(k, names[pid])].append(np.array(v))
for (k, cn), vs in sorted(rows.items()):
    vs = [v for v in vs if len(v) == len(vs[0])][2:]      # (drop the first dispatches: cold)
    if not vs:
        continue
    m = np.mean(vs, axis=0)
    tot = m.sum()
    n = len(m)
    print(f"{k[:60]:60s} {cn:16s} dispatches {len(vs):3d} instances {n:4d} sum/dispatch {tot:14.0f}  channel min/mean/max {m.min():10.0f} {m.mean():10.0f} {m.max():10.0f}"
          f"  (max/mean {m.max() / max(m.mean(), 1e-9):.3f}, cv {m.std() / max(m.mean(), 1e-9):.3f})")
    if n == 128:
        g16 = m.reshape(8, 16).sum(1)
        g8 = m.reshape(16, 8).sum(1)
        print(f"{'':60s} {'':16s} by 16 consecutive rows: " + " ".join(f"{x / tot * 8:.3f}" for x in g16))
        print(f"{'':60s} {'':16s} by row % 16          : " + " ".join(f"{x / tot * 16:.3f}" for x in m.reshape(8, 16).sum(0)))
