#!/usr/bin/env python
"""Same-box A/B of two builds of the library on the headline workload: alternates `bench.py --no-configs` runs between
CF_LIB_PATH=<a.so> and <b.so> and prints mean / standard deviation per build.  Box-to-box variation (35.2 .. 35.5 us per layer for one
binary) is larger than most single changes, so a change is only believed after an alternation like this one.

    python -m clusterfusion_amd.build --force && cp clusterfusion_amd/libclusterfusion_hip.so clusterfusion_amd/libexp_old.so
    CF_EXTRA_HIPCC_FLAGS=-DCF_EXP_X=1 python -m clusterfusion_amd.build --force && cp ... clusterfusion_amd/libexp_new.so
    gpurun -- python tools/ab_bench.py clusterfusion_amd/libexp_new.so clusterfusion_amd/libexp_old.so [rounds] [seq]
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seq = sys.argv[4] if len(sys.argv) > 4 else "4096"
res = {p: [] for p in libs}
for _ in range(rounds):
    for p in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-configs", "--seq", seq, "--steps", "200"],
                             env=dict(os.environ, CF_LIB_PATH=p), capture_output=True, text=True).stdout
        res[p].append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1])["us_per_layer"])
for p in libs:
    v = res[p]
    print(f"{os.path.basename(p):28s} n={len(v)} mean {statistics.mean(v):.3f} us  sd {statistics.stdev(v) if len(v) > 1 else 0:.3f}  {[round(x, 2) for x in v]}")
