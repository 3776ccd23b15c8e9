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
         dict(batch=8, S=0, lens=[8192] + [100] * 7), dict(batch=9, S=0, lens=[5, 0, 129, 1, 700, 0, 64, 2049, 3]),
         dict(batch=2, S=3000, hkv=8), dict(batch=4, S=8192, hkv=8), dict(batch=3, S=0, lens=[5000, 0, 700], hkv=8)]      # k_fused_decode_gb (round 6)


def main():
    g = torch.Generator(device=dev).manual_seed(9)
    total = 0
    for kw in CASES:
        layers = [config_bench.make_batch(g, kw["batch"], kw["S"], lens=kw.get("lens"), hkv=kw.get("hkv", 32)) if "batch" in kw else config_bench.make(g, **kw) for _ in range(6)]
        # (AUTO hands 30 .. 32 rows to the five launches since round 6: the persistent kernel still serves them when asked, and is what is soaked)
        cfa.set_path("fused" if kw.get("batch", 1) >= 30 else "auto")
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
        cfa.set_path("auto")
        print(f"{kw}: {n} launches, bit-identical, no exchange error")
        total += n
        del layers
    total += soak_tp_publish(g)
    print(f"soak ok: {total} persistent-kernel launches")


def soak_tp_publish(g, world=8):
    """The collective's publish folded into the shard kernel + the gather, `world` virtual ranks on this GPU: one graph = every rank's
    layer launch + every rank's gather, replayed for its share of the time; the reduced output must stay bit-identical (on every
    rank, every replay) and the receive areas' error words clear."""
    from clusterfusion_amd.tp import OneShotReducer
    n = 4096
    areas = [torch.zeros(OneShotReducer.area_bytes(world, n), dtype=torch.uint8, device=dev) for _ in range(world)]
    reds = [OneShotReducer(r, world, n, areas) for r in range(world)]
    base = config_bench.make(g, hidden=4096, hq=32 // world, hkv=32 // world, S=3000, layout="out_in", style="neox", residual=True)
    layers = [base.with_tp_publish(reds[0])] + [config_bench.make(g, hidden=4096, hq=32 // world, hkv=32 // world, S=3000, layout="out_in", style="neox",
                                                                  residual=True).with_tp_publish(reds[r]) for r in range(1, world)]
    outs = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(world)]

    def step():
        for p in layers:
            p.run()
        for r in range(world):
            reds[r].gather(outs[r])
    step()
    torch.cuda.synchronize()
    ref = outs[0].clone()
    assert all(torch.equal(o, ref) for o in outs)
    st = torch.cuda.Stream()
    count = 0
    with torch.cuda.stream(st):
        step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            step()
        t_end = time.time() + SECONDS / len(CASES)
        while time.time() < t_end:
            for _ in range(50):
                gr.replay()
            step()
            count += 51 * world
            torch.cuda.synchronize()
            assert all(torch.equal(o, ref) for o in outs), ("reduced output changed between repetitions", count)
    assert all(rd.error() == 0 for rd in reds)
    cfa.check_device_errors()
    print(f"TP publish in the shard kernel + gather, {world} virtual ranks: {count} layer launches (+ as many gathers), bit-identical on every rank, no error word")
    return count


main()
