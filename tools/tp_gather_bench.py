#!/usr/bin/env python
"""What the library's own all-reduce adds to a TP-8 rank's layer on ONE GPU with 8 virtual ranks (every rank's shard kernel publishes into
all 8 receive areas, every rank gathers): us per (8 layer launches + 8 gathers) / 8, for the separate gather launch and for the gather folded
into the fused add + RMSNorm, against the shard kernels alone."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import bench
import clusterfusion_amd as cfa
import config_bench
from clusterfusion_amd.tp import OneShotReducer

dev = torch.device("cuda:0")
world, n = 8, 4096
g = torch.Generator(device=dev).manual_seed(4)
areas = [torch.zeros(OneShotReducer.area_bytes(world, n), dtype=torch.uint8, device=dev) for _ in range(world)]
reds = [OneShotReducer(r, world, n, areas) for r in range(world)]
base = [config_bench.make(g, hidden=4096, hq=4, hkv=4, S=4096, layout="out_in", style="neox", residual=True) for _ in range(world)]
pub = [base[r].with_tp_publish(reds[r]) for r in range(world)]
outs = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(world)]
rw = (torch.randn(n, generator=g, device=dev) * 0.1).half()
res = (torch.randn(1, n, generator=g, device=dev) * 0.1).half()
nout = [torch.empty(1, n, dtype=torch.float16, device=dev) for _ in range(world)]
st = torch.cuda.Stream(dev)


def alone():
    for p in base:
        p.run()


def gather():
    for p in pub:
        p.run()
    for r in range(world):
        reds[r].gather(outs[r])


def norm_gather():
    for p in pub:
        p.run()
    for r in range(world):
        reds[r].gather_rmsnorm(rw, 1e-6, residual=res, out=nout[r])


def norm_plain():
    for p in base:
        o = p.run()[0]
    for r in range(world):
        cfa.rmsnorm(base[r].outputs[0], rw, 1e-6, residual=res, out=nout[r])


rec = {}
for name, fn in (("shard_alone", alone), ("publish_plus_gather", gather), ("publish_plus_norm_gather", norm_gather), ("shard_plus_plain_norm", norm_plain)):
    rec[name + "_us_per_rank"] = round(min(bench._graph_time_us(fn, world, 30, st) for _ in range(3)), 2)
rec["errors"] = [r.error() for r in reds]
gather()
torch.cuda.synchronize()
rec["identical_on_every_rank"] = all(torch.equal(outs[0], o) for o in outs)
cfa.check_device_errors()
print(json.dumps(rec))
