#!/usr/bin/env python
"""us per call of llama_decoder_layer_batch_decode_sglang over batch sizes (Llama-2-7B dims, paged KV,
page size 1, every row S cached tokens), NL distinct layers replayed from a hipGraph so that weights come from HBM.
Prints the algorithmic bytes (weights once + every row's K/V) and the fraction of the 8 TB/s roofline."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import clusterfusion_amd as cfa

dev = torch.device("cuda:0")
if os.environ.get("CF_FLAGS"):      # debug bits (32: 2 .. 4 rows through the stage pipeline instead of k_fused_decode_mhab)
    from clusterfusion_amd import _lib
    _lib.load().cf_debug_set_flags(int(os.environ["CF_FLAGS"]))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BATCHES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 32]
NL = int(os.environ.get("CF_NL", "32"))      # (a graph replay costs ~15 us of launch gap: amortised over NL calls)
H, HD = 4096, 128


def rn(g, *shape):
    return (torch.randn(*shape, generator=g, device=dev) * 0.1).half()


def main():
    g = torch.Generator(device=dev).manual_seed(3)
    wq = [rn(g, 3 * H, H) for _ in range(NL)]
    wo = [rn(g, H, H) for _ in range(NL)]
    rms = [rn(g, H) + 1 for _ in range(NL)]
    for bs in BATCHES:
        # CF_LENS="4000,300,..." : a ragged batch (one length per row, overrides S and the batch list); the pools are sized for the sum
        lens = [int(v) for v in os.environ["CF_LENS"].split(",")] if os.environ.get("CF_LENS") else [S] * bs
        bs = len(lens)
        n_slots = sum(lens) + bs
        kcs = [rn(g, n_slots, H) for _ in range(NL)]
        vcs = [rn(g, n_slots, H) for _ in range(NL)]
        kptrs = torch.tensor([t.data_ptr() for t in kcs], dtype=torch.uint64, device=dev)
        vptrs = torch.tensor([t.data_ptr() for t in vcs], dtype=torch.uint64, device=dev)
        perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(bs)).to(torch.int32).to(dev)
        indptr = torch.tensor([0] + [sum(lens[: i + 1]) + i + 1 for i in range(bs)], dtype=torch.int32, device=dev)
        positions = torch.tensor(lens, dtype=torch.int64, device=dev)
        cos_sin = (torch.rand(max(lens) + 1, 128, generator=g, device=dev) * 2 - 1).float()
        x, r = rn(g, bs, H), rn(g, bs, H)
        out, rout = torch.empty_like(x), torch.empty_like(x)

        def step():
            for l in range(NL):
                cfa.llama_decoder_layer_batch_decode_sglang(out, rout, x, r, wq[l], wo[l], indptr, perm, kptrs, vptrs, l,
                                                            rms[l], 1e-6, positions, cos_sin)
        step()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            step()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                step()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                gr.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (reps * NL) * 1e6
        cfa.profile_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        stage_ms, ncalls = cfa.profile_read(reset=True)
        cfa.profile_enable(False)
        stages = [round(m * 1e3 / max(ncalls, 1), 1) for m in stage_ms]
        byt = 2 * H * 3 * H + 2 * H * H + sum(lens) * 4 * H
        print(json.dumps({"batch": bs, "S": S if not os.environ.get("CF_LENS") else lens, "path": cfa.last_path(), "kernel": cfa.last_variant(), "us_per_call": round(us, 2), "MB": round(byt / 1e6, 1),
                          "frac_of_8TBs": round(byt / us / 1e3 / 8000, 3), "us_per_row": round(us / bs, 2),
                          "stage_us_events(qkv,attn,oproj,-)": stages}))
        del kcs, vcs


main()
