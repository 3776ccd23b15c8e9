"""us per call of the plain entry's native [in,out] kernel (weight re-layout off; GPT-J rotation, contiguous cache), 32 distinct layer states per
graph replay:  CF_LIB_PATH=... python tools/plain_ab.py S   (the [out,in] kernels: tools/shard_ab.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench, config_bench
import clusterfusion_amd as cfa
S = int(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
cfa.set_weight_relayout(False)
ls = [config_bench.make(g, hidden=4096, hq=32, hkv=32, S=S, layout="in_out", style="gptj", residual=False) for _ in range(32)]
st = torch.cuda.Stream(dev)
us = [bench._graph_time_us(lambda: [p.run() for p in ls], len(ls), 30, st) for _ in range(3)]
cfa.check_device_errors()
print(f"in_out S={S} {cfa.last_variant()}: {min(us):.2f} us (runs {[round(u, 2) for u in us]})")
