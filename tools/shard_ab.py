#!/usr/bin/env python
"""us per call of one shard geometry, 32 distinct layer states per graph replay (HBM, not the Infinity Cache):
    CF_LIB_PATH=... [CF_DEBUG_FLAGS=bits] python tools/shard_ab.py hq hkv S [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import bench
import clusterfusion_amd as cfa
import config_bench

hq, hkv, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if os.environ.get("CF_DEBUG_FLAGS"):      # experiment bits of the library (cf_debug_set_flags), e.g. 256: k_fused_decode_s<8> for 8q/8kv
    from clusterfusion_amd import _lib
    _lib.load().cf_debug_set_flags(int(os.environ["CF_DEBUG_FLAGS"]))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
ls = [config_bench.make(g, hidden=4096, hq=hq, hkv=hkv, S=S, layout="out_in", style="neox", residual=True) for _ in range(32)]
st = torch.cuda.Stream(dev)
us = [bench._graph_time_us(lambda: [p.run() for p in ls], len(ls), 30, st) for _ in range(3)]
cfa.check_device_errors()
print(f"{hq}q/{hkv}kv S={S} {cfa.last_variant()}: {min(us):.2f} us (runs {[round(u, 2) for u in us]})")
