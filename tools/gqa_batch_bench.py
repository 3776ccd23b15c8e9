#!/usr/bin/env python
"""us per call of the grouped-query model (32q/8kv, Llama-3-8B shapes) with 2 .. 4 sequences: k_fused_decode_gb (one persistent launch)
against the five launches (debug flag 32), 32 distinct layers per graph replay.
    python tools/gqa_batch_bench.py S [rows,...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import clusterfusion_amd as cfa
from clusterfusion_amd import _lib

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ROWS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 4]
NL, H, HD = 32, 4096, 128
g = torch.Generator(device=dev).manual_seed(9)


def rn(*shape):
    return (torch.randn(*shape, generator=g, device=dev) * 0.1).half()


wq = [rn(48 * HD, H) for _ in range(NL)]
wo = [rn(H, H) for _ in range(NL)]
rms = [rn(H) for _ in range(NL)]
st = torch.cuda.Stream(dev)
for bs in ROWS:
    n_slots = bs * (S + 1)
    kcs = [rn(n_slots, 8 * HD) for _ in range(NL)]
    vcs = [rn(n_slots, 8 * HD) for _ in range(NL)]
    perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(bs)).to(torch.int32).to(dev)
    indptr = (torch.arange(bs + 1, dtype=torch.int32) * (S + 1)).to(dev)
    pos = torch.full((bs,), S, dtype=torch.int64, device=dev)
    lens = torch.full((bs,), S, dtype=torch.int32, device=dev)
    cs = (torch.rand(S + 1, 128, generator=g, device=dev) * 2 - 1).float()
    x, r = rn(bs, H), rn(bs, H)
    ls = [cfa.prepare_decoder_layer(x, r, wq[l], wo[l], kcs[l], vcs[l], rms[l], 1e-6, cs, cs.view(-1)[64:], n_q_heads=32, n_kv_heads=8,
                                    kv_indptr=indptr, kv_indices=perm, kv_seq_lens=lens, page_size=1, positions=pos, rope_row_stride=128,
                                    write_kv_to_cache=True, max_seq_len=S) for l in range(NL)]
    b = 2 * H * 48 * HD + 2 * H * H + bs * 4 * S * 8 * HD
    rec = {"rows": bs, "S": S, "MB": round(b / 1e6, 1)}
    for name, flag in (("kernel", 0), ("five_launches", 32)):
        _lib.load().cf_debug_set_flags(flag)
        us = min(bench._graph_time_us(lambda: [p.run() for p in ls], NL, 20, st) for _ in range(2))
        rec[name + "_us"] = round(us, 2)
        rec[name] = cfa.last_variant()
        rec[name + "_frac"] = round(b / (us * 1e-6) / 8e12, 3)
    _lib.load().cf_debug_set_flags(0)
    print(json.dumps(rec), flush=True)
    del kcs, vcs, ls
cfa.check_device_errors()
