#!/usr/bin/env python
"""Whole-model greedy decode tokens/s with the fused op in its real place: Llama-2-7B shapes,
random weights, S cached tokens, one HIP graph per token (attention block = the fused op; FFN / LM head = torch).
Prints one JSON line; also the share of a token spent in the fused op (measured separately)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import clusterfusion_amd as cfa
from clusterfusion_amd.harness import DecodeModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
GQA = len(sys.argv) > 3 and sys.argv[3] == "llama3"


def main():
    kw = dict(n_kv_heads=8, ffn=14336, vocab=128256) if GQA else {}
    m = DecodeModel(start_pos=S, max_seq=S + STEPS * 3 + 64, **kw)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            m.step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            m.step()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            gr.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    err = None
    try:
        cfa.check_device_errors()
    except Exception as e:   # noqa: BLE001
        err = str(e)
    ms = dt / STEPS * 1e3
    print(json.dumps({"model": "Llama-3-8B shapes" if GQA else "Llama-2-7B shapes", "S_start": S, "steps": STEPS,
                      "ms_per_token": round(ms, 3), "tok_s": round(1e3 / ms, 1), "path": cfa.last_path(),
                      "final_pos": int(m.pos.item()), "device_error": err,
                      "note": "attention block = fused op; RMSNorms = clusterfusion.rmsnorm; FFN, LM head, argmax = torch"}))


main()
