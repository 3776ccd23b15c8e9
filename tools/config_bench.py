#!/usr/bin/env python
"""Per-config timing of the op on ONE GPU (BASELINE.json configs 2-5): N distinct layer states are
cycled so reads come from HBM; reports us/call and fraction of the 8 TB/s HBM peak."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import clusterfusion_amd as cfa

dev = torch.device("cuda:0")
if os.environ.get("CF_FLAGS"):      # debug bits (cf_debug_set_flags), e.g. 128: the generic kernel for the TP-8 shard
    from clusterfusion_amd import _lib as _cf_lib
    _cf_lib.load().cf_debug_set_flags(int(os.environ["CF_FLAGS"]))


def rn(g, *shape):
    return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.1).half()


def make(g, hidden, hq, hkv, S, layout, style, residual):
    qd, kd = hq * 128, hkv * 128
    x = rn(g, 1, hidden)
    res = rn(g, 1, hidden) if residual else None
    if layout == "out_in":
        w_qkv, w_o = rn(g, qd + 2 * kd, hidden), rn(g, hidden, qd)
    else:
        w_qkv, w_o = rn(g, 3 * hidden, qd), rn(g, qd, hidden)
    kc, vc = rn(g, S, kd), rn(g, S, kd)
    n = 64 if style == "neox" else 128
    ang = torch.rand(n, generator=g, device=dev) * 6.28
    return cfa.prepare_decoder_layer(x, res, w_qkv, w_o, kc, vc, rn(g, hidden), 1e-6, ang.cos(), ang.sin(),
                                     n_q_heads=hq, n_kv_heads=hkv, weight_layout=layout, rope_style=style, want_kv=True)


def make_batch(g, bs, S, page_size=1, lens=None, hkv=32):
    """bs sequences of S cached tokens each (or `lens`: one length per row), paged KV (scattered slots), Llama-2-7B dims (hkv=8: the
    32q/8kv geometry of Llama-3-8B), [out,in] weights."""
    H = 4096
    if lens is not None:
        return _make_ragged(g, list(lens), hkv)
    n_slots = bs * (S + page_size)
    x, res = rn(g, bs, H), rn(g, bs, H)
    w_qkv, w_o = rn(g, (32 + 2 * hkv) * 128, H), rn(g, H, H)
    kc, vc = rn(g, n_slots, hkv * 128), rn(g, n_slots, hkv * 128)
    sh = int(os.environ.get("CF_KV_SHIFT", "0"))      # experiment: the pools start `sh` fp16 elements into their allocation
    if sh:
        kc = torch.cat([kc.view(-1), kc.view(-1)[:sh]])[sh:].view(n_slots, hkv * 128)
        vc = torch.cat([vc.view(-1), vc.view(-1)[:sh]])[sh:].view(n_slots, hkv * 128)
    per = (S + 1 + page_size - 1) // page_size
    perm = torch.randperm(n_slots // page_size, generator=torch.Generator().manual_seed(bs))[: bs * per].to(torch.int32).to(dev)
    indptr = (torch.arange(bs + 1, dtype=torch.int32) * per).to(dev)
    positions = torch.full((bs,), S, dtype=torch.int64, device=dev)
    cos_sin = (torch.rand(S + 1, 128, generator=g, device=dev) * 2 - 1).float()
    return cfa.prepare_decoder_layer(x, res, w_qkv, w_o, kc, vc, rn(g, H), 1e-6, cos_sin, cos_sin.view(-1)[64:], n_q_heads=32, n_kv_heads=hkv,
                                     kv_indptr=indptr, kv_indices=perm, kv_seq_lens=positions.to(torch.int32), page_size=page_size,
                                     positions=positions, rope_row_stride=128, write_kv_to_cache=True, max_seq_len=S, want_kv=False)


def _make_ragged(g, lens, hkv=32):
    """page size 1, one length per row (k_fused_decode_mhaq's records / deferred merges only run on ragged batches)."""
    H, bs = 4096, len(lens)
    n_slots = sum(lens) + bs
    x, res = rn(g, bs, H), rn(g, bs, H)
    w_qkv, w_o = rn(g, (32 + 2 * hkv) * 128, H), rn(g, H, H)
    kc, vc = rn(g, n_slots, hkv * 128), rn(g, n_slots, hkv * 128)
    perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(bs)).to(torch.int32).to(dev)
    indptr = torch.tensor([0] + [sum(lens[: i + 1]) + i + 1 for i in range(bs)], dtype=torch.int32, device=dev)
    positions = torch.tensor(lens, dtype=torch.int64, device=dev)
    cos_sin = (torch.rand(max(lens) + 1, 128, generator=g, device=dev) * 2 - 1).float()
    return cfa.prepare_decoder_layer(x, res, w_qkv, w_o, kc, vc, rn(g, H), 1e-6, cos_sin, cos_sin.view(-1)[64:], n_q_heads=32, n_kv_heads=hkv,
                                     kv_indptr=indptr, kv_indices=perm, kv_seq_lens=positions.to(torch.int32), page_size=1,
                                     positions=positions, rope_row_stride=128, write_kv_to_cache=True, max_seq_len=max(lens), want_kv=False)


def run(name, nlayers=12, reps=30, **kw):
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [make(g, **kw) for _ in range(nlayers)]
    torch.cuda.synchronize()
    for p in layers:
        p.run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for p in layers:
                p.run()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    us = dt / (reps * nlayers) * 1e6
    b = cfa.algorithmic_bytes(kw["S"], kw["hidden"], kw["hq"], kw["hkv"], 128, 1, kw["residual"])
    print(json.dumps({"config": name, "path": cfa.last_path(), "us_per_call": round(us, 2), "MB": round(b / 1e6, 1),
                      "GB_s": round(b / us / 1e3, 0), "frac_of_8TBs": round(b / us / 1e3 / 8000, 3)}))


if __name__ == "__main__" and "--gqaonly" in sys.argv:
    run("4: Llama-3-8B GQA 32/8 S=8192", nlayers=32, hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
    run("TP-2 shard (16 heads) S=4096", nlayers=32, hidden=4096, hq=16, hkv=16, S=4096, layout="out_in", style="neox", residual=True)
    sys.exit(0)

if __name__ == "__main__" and "--shards" in sys.argv:      # config 4 and the three shard geometries, 32 layers each
    run("4: Llama-3-8B GQA 32/8 S=8192", nlayers=32, hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
    for tp, hq in ((8, 4), (4, 8), (2, 16)):
        run(f"5: TP={tp} shard ({hq} heads) S=4096", nlayers=32, hidden=4096, hq=hq, hkv=hq, S=4096, layout="out_in", style="neox", residual=True)
    sys.exit(0)

if __name__ == "__main__" and "--tp8only" in sys.argv:
    run("5: Llama-2-7B TP=8 shard (4 heads) S=4096, local compute only", nlayers=32, hidden=4096, hq=4, hkv=4, S=4096, layout="out_in", style="neox", residual=True)
    sys.exit(0)

if __name__ == "__main__" and "--gqa" not in sys.argv and "--stages" not in sys.argv and "--none" not in sys.argv:
    run("2: Llama-2-7B plain API ([in,out], GPT-J) S=1024", hidden=4096, hq=32, hkv=32, S=1024, layout="in_out", style="gptj", residual=False)
    run("2b: Llama-2-7B sglang ([out,in], NEOX) S=1024", hidden=4096, hq=32, hkv=32, S=1024, layout="out_in", style="neox", residual=True)
    run("3: Llama-2-7B sglang S=4096 contiguous", hidden=4096, hq=32, hkv=32, S=4096, layout="out_in", style="neox", residual=True)
    run("3b: Llama-2-7B plain API S=4096", hidden=4096, hq=32, hkv=32, S=4096, layout="in_out", style="gptj", residual=False)
    run("4: Llama-3-8B GQA 32/8 S=8192", hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
    run("5: Llama-2-7B TP=8 shard (4 heads) S=4096, local compute only", nlayers=32, hidden=4096, hq=4, hkv=4, S=4096, layout="out_in", style="neox", residual=True)
    run("S=128 sglang", hidden=4096, hq=32, hkv=32, S=128, layout="out_in", style="neox", residual=True)


def stages(name, **kw):
    g = torch.Generator(device=dev).manual_seed(1)
    layers = [make(g, **kw) for _ in range(8)]
    torch.cuda.synchronize()
    for p in layers:
        p.run()
    cfa.profile_enable(True)
    for _ in range(10):
        for p in layers:
            p.run()
    torch.cuda.synchronize()
    ms, n = cfa.profile_read()
    cfa.profile_enable(False)
    print(name, "stage us (events):", [round(m * 1e3 / n, 2) for m in ms])


if __name__ == "__main__" and "--stages" in sys.argv:
    stages("4 GQA S=8192", hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
    stages("5 TP8 shard", hidden=4096, hq=4, hkv=4, S=4096, layout="out_in", style="neox", residual=True)


if __name__ == "__main__" and "--gqa" in sys.argv:
    for ns in (0, 8, 16, 32, 64):
        cfa.set_tuning(ns)
        run(f"4: GQA S=8192 kv_splits={ns}", hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
        stages(f"   splits={ns}", hidden=4096, hq=32, hkv=8, S=8192, layout="out_in", style="neox", residual=True)
    cfa.set_tuning(0)
