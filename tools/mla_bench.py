"""Timing of the DeepSeek MLA decoder-layer op (cf_deepseek_decoder_layer) on one MI355X.

    python tools/mla_bench.py [--seq 4096] [--layers 27] [--steps 50] [--rope-scores]

A "step" = one new token through the op of `layers` distinct layers (DeepSeek-V2-Lite has 27; distinct weights and
caches per layer, so nothing is served from a warm cache).  Prints one JSON line: us / layer (HIP events around the
timed region), the roofline fraction of the whole op against 8 TB/s on its algorithmic bytes, and the per-stage
split (separate, synchronising pass).  The CPU oracle is only the checker (first layer, before timing)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clusterfusion_amd as cfa  # noqa: E402
from oracle import mla_oracle as M  # noqa: E402

ORDER = ["input", "weight_q_nope", "weight_q_pe", "weight_uk", "weight_kv_nope", "weight_k_pe", "weight_uv", "weight_o",
         "ckv_cache", "rms_input_weight", "rms_ckv_weight", "cos", "sin"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=27)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rope-scores", action="store_true")
    ap.add_argument("--path", default="auto", choices=["auto", "pipeline", "fused"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfa.set_path(a.path)
    layers = []
    for li in range(a.layers):
        inp = M.make_mla_inputs(1000 + li, a.seq, score_gain=3.0)
        layers.append([inp[k].to(dev) for k in ORDER])
        if li == 0:
            ref = M.mla_decoder_layer(inp, rope_scores=a.rope_scores)["o"]
            o = cfa.deepseek_decoder_layer(*layers[0], rope_scores=a.rope_scores)
            err = (o.cpu().double() - ref).abs().max().item()
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err

    def step():
        for g in layers:
            cfa.deepseek_decoder_layer(*g, rope_scores=a.rope_scores)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    us_layer = e0.elapsed_time(e1) * 1e3 / (a.steps * a.layers)
    # CUDA-graph replay of one step: what a serving loop would launch
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us_layer_graph = e0.elapsed_time(e1) * 1e3 / (a.steps * a.layers)
    cfa.deepseek_profile(True)
    cfa.deepseek_profile(None)
    for _ in range(5):
        step()
    ms, n = cfa.deepseek_profile(None)
    cfa.deepseek_profile(False)
    cfa.check_device_errors()
    nbytes = cfa.deepseek_algorithmic_bytes(a.seq, a.rope_scores)
    best = min(us_layer, us_layer_graph)
    print(json.dumps({
        "op": "deepseek_decoder_layer", "seq_len": a.seq, "layers": a.layers, "steps": a.steps,
        "rope_scores": a.rope_scores, "path": cfa.last_path(), "us_per_layer": round(us_layer, 2), "us_per_layer_graph": round(us_layer_graph, 2),
        "algorithmic_bytes": nbytes,
        "roofline": {"bound": "hbm", "achieved": round(nbytes / best / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(nbytes / best / 1e3 / 8000.0, 4)},
        "stage_us": {k: round(v * 1e3 / n, 2) for k, v in zip(["proj_absorb", "attn", "uv_out"], ms)},
    }))


if __name__ == "__main__":
    main()
