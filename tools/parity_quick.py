"""Quick parity of persistent kernels of the library under CF_LIB_PATH against the oracle (development: run an experimental
build through this before timing it with tools/ab_libs.sh):  python tools/parity_quick.py [hq,hkv ...]   (default 32,8 16,4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import clusterfusion_amd as cfa
from oracle import cf_oracle as O
DEV = torch.device("cuda:0")
cfa.set_path("fused")
if os.environ.get("CF_DEBUG_FLAGS"):      # experiment bits of the library (cf_debug_set_flags)
    from clusterfusion_amd import _lib
    _lib.load().cf_debug_set_flags(int(os.environ["CF_DEBUG_FLAGS"]))
worst = 0
GEOMS = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(32, 8), (16, 4)]
for hq, hkv in GEOMS:
    dims = O.LayerDims(4096, hq, hkv, 128)
    for S in [0, 100, 257, 1000, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 9000]:
        inp = O.make_inputs(700 + S, S, dims)
        g = {k: v.to(DEV) for k, v in inp.items()}
        ro, rr, rk, rv = O.decoder_layer(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                         inp["rms_w"], 1e-5, inp["cos"], inp["sin"], dims=dims)
        res = g["residual"].clone()
        o, r, k, v = cfa.decoder_layer(g["x"], res, g["weight_qkv"], g["weight_o"], g["k_cache"], g["v_cache"], g["rms_w"], 1e-5,
                                       g["cos"], g["sin"], n_q_heads=hq, n_kv_heads=hkv, residual_out=res)
        torch.cuda.synchronize()
        cfa.check_device_errors()
        e = (o.cpu().float() - ro.float()).abs().max().item() / max(1.0, ro.float().abs().max().item())
        print(f"{hq}/{hkv} S={S:6d} {cfa.last_variant():24s} rel err {e:.2e} residual {torch.equal(r.cpu(), rr)}")
        worst = max(worst, e)
print("WORST", worst)
