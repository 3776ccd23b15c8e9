#!/usr/bin/env python
"""Iterative tuning of the phase-1 share table of the persistent [out,in] MHA kernel (cf_api.hip fill_p1_shares).

Each round: (1) in-kernel stamps of the bench workload under hipGraph replay (tools/fused_timeline.py) give the median
absolute time at which every workgroup finishes phase 2; (2) per cell (b / 64, b % 8) the share moves by
(mean - cell) / us_per_pair, clipped to +-2 pairs; (3) bench.py measures the table.  Prints every table tried.

    python tools/tune_p1_shares.py [rounds]
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
US_PER_PAIR = 0.83


def run_bench(env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-configs", "--steps", "100"], env=env, capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    return d["us_per_layer"]


def main():
    table = np.array([[e if x % 2 == 0 else o for x in range(8)] for e, o in ((30, 23), (22, 15), (30, 23), (28, 21))])   # cf_api.hip P1_SHARE
    os.makedirs(os.path.join(ROOT, "gpurun_out", "tune"), exist_ok=True)
    for it in range(ROUNDS + 1):
        env = dict(os.environ, CF_P1_TABLE=",".join(str(v) for v in table.flatten()))
        us = [run_bench(env) for _ in range(2)]
        print(f"round {it}: table {table.flatten().tolist()}  bench us/layer {us}", flush=True)
        if it == ROUNDS:
            break
        npy = os.path.join(ROOT, "gpurun_out", "tune", f"abs{it}.npy")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_timeline.py"), "4096", "0"],
                       env=dict(env, CF_TL_GRAPH="1", CF_TL_ABS="1", CF_TL_SAVE=npy), capture_output=True, text=True)
        p2 = np.load(npy)[:, 3]
        b = np.arange(256)
        cell = np.array([[np.median(p2[((b >> 6) == s) & ((b & 7) == x)]) for x in range(8)] for s in range(4)])
        print("   cell medians of 'P2 done':", np.round(cell, 2).tolist(), flush=True)
        adj = np.clip(np.round((cell.mean() - cell) / US_PER_PAIR), -2, 2).astype(int)
        new = np.clip(table + adj, 8, 32)
        # keep the sum: spread the difference over the cells whose unrounded correction was cut the most
        diff = 768 - int(new.sum())
        order = np.argsort(-(cell.mean() - cell).flatten()) if diff > 0 else np.argsort((cell.mean() - cell).flatten())
        flat = new.flatten()
        i = 0
        while diff != 0:
            k = order[i % 32]
            if 8 <= flat[k] + np.sign(diff) <= 32:
                flat[k] += np.sign(diff)
                diff -= np.sign(diff)
            i += 1
        table = flat.reshape(4, 8)


if __name__ == "__main__":
    main()
