#!/usr/bin/env python
"""Soak run of the persistent kernels: many thousands of back-to-back launches (graph replay and eager, all fused
specialisations, several sequence lengths); afterwards the sticky exchange-error word must be clear and every
repetition of the same inputs must have produced bit-identical outputs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import clusterfusion_amd as cfa
import config_bench

dev = torch.device("cuda:0")
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
CASES = [dict(hidden=4096, hq=32, hkv=32, S=4096, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=32, hkv=32, S=777, layout="in_out", style="gptj", residual=False),
         dict(hidden=4096, hq=32, hkv=32, S=9000, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=4, hkv=4, S=4096, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=16, hkv=16, S=300, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=8, hkv=8, S=2000, layout="out_in", style="neox", residual=True),
         dict(batch=2, S=1024), dict(batch=3, S=600), dict(batch=4, S=1500),      # small-batch kernels (paged, 2 / 4 row slots)
         dict(batch=5, S=700), dict(batch=8, S=1024), dict(batch=13, S=333), dict(batch=16, S=1024),      # k_fused_decode_mhaq<1>
         dict(batch=17, S=500), dict(batch=32, S=300),                                               # k_fused_decode_mhaq<2>
         dict(hidden=4096, hq=16, hkv=4, S=3000, layout="out_in", style="neox", residual=True),      # GQA shards (round 4)
         dict(hidden=4096, hq=8, hkv=2, S=5000, layout="out_in", style="neox", residual=True),
         dict(hidden=4096, hq=4, hkv=1, S=9000, layout="out_in", style="neox", residual=True),
         dict(batch=8, S=0, lens=[4000, 300, 1200, 50, 2500, 800, 100, 3000]),      # ... rows spanning token ranges: records, deferred merges
         dict(batch=8, S=0, lens=[8192] + [100] * 7), dict(batch=9, S=0, lens=[5, 0, 129, 1, 700, 0, 64, 2049, 3])]


def main():
    g = torch.Generator(device=dev).manual_seed(9)
    total = 0
    for kw in CASES:
        layers = [config_bench.make_batch(g, kw["batch"], kw["S"], lens=kw.get("lens")) if "batch" in kw else config_bench.make(g, **kw) for _ in range(6)]
        for p in layers:
            p.run()
        torch.cuda.synchronize()
        assert cfa.last_path() == "fused", kw
        ref = [p.outputs[0].clone() for p in layers]
        st = torch.cuda.Stream()
        n = 0
        with torch.cuda.stream(st):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for p in layers:
                    p.run()
            t_end = time.time() + SECONDS / len(CASES)
            while time.time() < t_end:
                for _ in range(50):
                    gr.replay()
                for p in layers:            # a few eager launches in between
                    p.run()
                n += 51 * len(layers)
                torch.cuda.synchronize()
                for p, r in zip(layers, ref):
                    assert torch.equal(p.outputs[0], r), ("output changed between repetitions", kw, n)
        cfa.check_device_errors()
        print(f"{kw}: {n} launches, bit-identical, no exchange error")
        total += n
        del layers
    print(f"soak ok: {total} persistent-kernel launches")


main()
