#!/usr/bin/env python
"""Same-process A/B of two debug-flag settings of the persistent kernel on the headline workload (S paged, 32 distinct layers,
hipGraph replay): checks that both produce bit-identical outputs, then alternates timed replays.

    python tools/ab_flags.py FLAGS_A FLAGS_B [seq] [rounds]       e.g.  0 512  (static shares vs ticketed phase 1)
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import clusterfusion_amd as cfa
from clusterfusion_amd import _lib

fa, fb = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
lib = _lib.load()
layers = bench.build_layers(cfa, dev, 1, 0, 32, S, 16)[0]
cfa.set_path("fused")
stream = torch.cuda.Stream(dev)
graphs, outs = {}, {}
with torch.cuda.stream(stream):
    for f in (fa, fb):
        lib.cf_debug_set_flags(f)
        for p in layers:
            p.run()
        torch.cuda.synchronize()
        cfa.check_device_errors()
        outs[f] = [torch.cat([o.clone().view(-1).view(torch.int16) for o in p.outputs if o is not None]) for p in layers]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for p in layers:
                p.run()
        graphs[f] = g
    same = all(torch.equal(a, b) for a, b in zip(outs[fa], outs[fb]))
    print(f"outputs of flags {fa} and {fb} bit-identical over 32 chained layers: {same}")
    res = {fa: [], fb: []}
    for r in range(rounds):
        for f in ((fa, fb) if r % 2 == 0 else (fb, fa)):
            g = graphs[f]
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(100):
                g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) * 1e3 / (100 * len(layers)))
    cfa.check_device_errors()
    # replays again must reproduce the outputs (tickets are per-launch state: nothing may leak from one launch to the next)
    for f in (fa, fb):
        graphs[f].replay()
        torch.cuda.synchronize()
        again = [torch.cat([o.view(-1).view(torch.int16) for o in p.outputs if o is not None]) for p in layers]
        print(f"flags {f}: replay reproduces the eager outputs: {all(torch.equal(a, b) for a, b in zip(again, outs[f]))}")
for f in (fa, fb):
    v = res[f]
    print(f"flags {f:4d} S={S}: mean {statistics.mean(v):.3f} us/layer  sd {statistics.stdev(v):.3f}  {[round(x, 2) for x in v]}")
