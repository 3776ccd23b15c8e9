#!/usr/bin/env python
"""bench.py -- headline measurement of the fused decoder-layer decode path on MI355X.

Metric (BASELINE.json): us / decoder-layer + decode tok/s, Llama-2-7B bs=1 seq=4096; % HBM roofline.

A "step" = ONE decoded token through the fused attention-block op of all 32 layers of Llama-2-7B
(32 calls of the hot path, each with ITS OWN weights and paged KV cache -- 6.4 GB of distinct bytes
per step, so the 256 MiB Infinity Cache cannot serve the reads, exactly as in a real decode).
Workload at N=1 = BASELINE configs[2]: bs=1, seq=4096, paged KV (page_size 16, pages scattered over
a pool 2x the needed size), [out,in] weights, NEOX RoPE, residual add, new K/V written to the cache.
N>1 = head-parallel TP over N GPUs (configs[4]): each rank holds 32/N heads of every layer and ONE
RCCL all-reduce of the 8 KB fp16 O-projection partial closes each layer; total work is fixed, so
scaling is "strong".  Inputs are synthetic (seeded randn*0.1 drawn on the device, random-init
weights), resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without WORLD_SIZE: spawns its N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: every rank slices ITS heads out of the same seeded full model, so the sharded path is checkable: outside the timed
region one full layer runs through the 1-GPU kernel on rank 0 and through shard + all-reduce on all ranks (`tp_parity`:
max-abs <= 2e-3, identical bits on every rank).  The timed region runs with RCCL's all-reduce (the headline, as north_star
asks) and again with the library's one-shot all-reduce over peer-mapped buffers (`oneshot`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HIDDEN, HEADS, HEAD_DIM, LAYERS = 4096, 32, 128, 32
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md); ~6290 measured copy
PREWARM = 40            # untimed steps in front of the caller's warm-up (clock ramp; see timed_leg)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--page-size", type=int, default=16)
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (2, 4, 5-shard)")
    ap.add_argument("--kv-splits", type=int, default=0)
    ap.add_argument("--path", default="auto", choices=["auto", "pipeline", "fused"])
    ap.add_argument("--debug-flags", type=int, default=0, help="experiment bits for the fused kernel (cf_debug_set_flags)")
    ap.add_argument("--only-eager", action="store_true", help="print only the eager drop-in-call figures (eager_entries) and exit")
    ap.add_argument("--spawn", action="store_true", help="re-launch through torch.distributed.run even for --gpus 1")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch check without a GPU: spawn the ranks, rendezvous over gloo, one all-reduce, one JSON line")
    ap.add_argument("--no-oneshot", action="store_true", help="N > 1: skip the leg with the library's one-shot all-reduce")
    ap.add_argument("--leg-timeout", type=float, default=float(os.environ.get("CF_BENCH_LEG_TIMEOUT", "240")),
                    help="N > 1: seconds the optional legs (one-shot all-reduce, in-kernel publish) get before rank 0 prints "
                         "the line without them")
    return ap.parse_args()


def build_layers(cfa, dev, world, rank, n_layers, S, page_size, seed=42, keep_full=0, extra_ranks=()):
    """Per-layer synthetic state on the device + one PreparedLayer per layer.

    world > 1: rank `rank` of a head-parallel shard.  Every rank draws the SAME full layer (same seed, same generator stream)
    and keeps its heads' slice (clusterfusion_amd.tp.shard_layer_weights / shard_kv_cache), so the ranks together hold one
    model and the reduced output can be checked against the unsharded kernel.  `keep_full` = how many leading layers also
    return their full tensors (rank 0's side of `tp_parity`); `extra_ranks` = further shards of those layers to build here
    (a process that plays several ranks of the shard in turn: CF_BENCH_TP > WORLD_SIZE, the one-GPU test of this path).
    world == 1: the slice is the whole layer -- bit-identical to what rounds 1-3 measured on."""
    from clusterfusion_amd.tp import ShardSpec, shard_kv_cache, shard_layer_weights
    g = torch.Generator(device=dev).manual_seed(seed)
    hq = HEADS // world
    qd = HEADS * HEAD_DIM

    def rn(*shape):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.1).half()

    n_pages_need = (S + 1 + page_size - 1) // page_size
    pool_pages = 2 * n_pages_need
    gx = torch.Generator(device=dev).manual_seed(seed)      # replicated tensors: same on every rank
    x0 = (torch.randn(1, HIDDEN, generator=gx, device=dev) * 0.1).half()
    res0 = (torch.randn(1, HIDDEN, generator=gx, device=dev) * 0.1).half()
    ang = torch.rand(HEAD_DIM // 2, generator=gx, device=dev) * (2 * math.pi)
    cos_sin = torch.cat([ang.cos(), ang.sin()]).repeat(S + 8, 1).contiguous()       # [max_pos, 128]
    positions = torch.tensor([S], dtype=torch.int64, device=dev)
    seq_lens = torch.tensor([S], dtype=torch.int32, device=dev)
    indptr = torch.tensor([0, n_pages_need], dtype=torch.int32, device=dev)

    def prepare(x, res, w_qkv, w_o, rms_w, kc, vc, perm, heads):
        out = torch.empty(1, HIDDEN, dtype=torch.float16, device=dev)
        res_out = torch.empty(1, HIDDEN, dtype=torch.float16, device=dev)
        return cfa.prepare_decoder_layer(
            x, res, w_qkv, w_o, kc, vc, rms_w, 1e-6, cos_sin, cos_sin.view(-1)[HEAD_DIM // 2:],
            n_q_heads=heads, n_kv_heads=heads, kv_indptr=indptr, kv_indices=perm, kv_seq_lens=seq_lens,
            page_size=page_size, max_seq_len=S, positions=positions, rope_row_stride=HEAD_DIM,
            out=out, residual_out=res_out, write_kv_to_cache=True, want_kv=False)

    def shard(r, w_qkv, w_o, kc, vc):
        if world == 1:
            return w_qkv, w_o, kc, vc
        spec = ShardSpec(HIDDEN, HEADS, HEADS, HEAD_DIM, r, world)
        ws, wos = shard_layer_weights(w_qkv, w_o, spec)
        return ws, wos, shard_kv_cache(kc, spec), shard_kv_cache(vc, spec)

    layers, full, extra = [], [], []
    x, res = x0, res0
    for li in range(n_layers):
        w_qkv = rn(3 * qd, HIDDEN)
        w_o = rn(HIDDEN, qd)
        rms_w = rn(HIDDEN)
        kc = rn(pool_pages * page_size, qd)
        vc = rn(pool_pages * page_size, qd)
        perm = torch.randperm(pool_pages, generator=g, device=dev)[:n_pages_need].to(torch.int32).contiguous()
        ws, wos, kcs, vcs = shard(rank, w_qkv, w_o, kc, vc)
        p = prepare(x, res, ws, wos, rms_w, kcs, vcs, perm, hq)
        layers.append(p)
        if li < keep_full and world > 1:
            full.append(prepare(x, res, w_qkv, w_o, rms_w, kc, vc, perm, HEADS))
            others = []
            for r in extra_ranks:
                ws, wos, kcs, vcs = shard(r, w_qkv, w_o, kc, vc)
                others.append(prepare(x, res, ws, wos, rms_w, kcs, vcs, perm, hq))
            extra.append(others)
        # chain like a model: next layer's input = this layer's output, residual = updated residual
        # (the FFN half between them is outside the fused op: chat/llama/model.py:519)
        x, res = p.outputs[0], p.outputs[1]
    return layers, full, extra


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(S_headline, budget_s=24.0):
    """The reference's PyTorch-eager layer (tests/test_llama.py:57-113 == tests/test_llama_tilelang.py:18-49), restated in
    oracle/cf_oracle.py, timed on THIS box's host cores: BASELINE config 1 / BASELINE.md section 3 -- Llama-2-7B single
    layer, bs=1, S=128, seed 42, all tensors randn*0.1 -- in two variants (fp16 weights through half matmuls, as the
    reference's nn.Linear layers run; weights pre-converted to fp32), 3 warm-ups, median of up to 30 calls each, plus the
    S of the headline workload (fp32) for scale.  Reported baseline, not a target."""
    from oracle import cf_oracle as O

    def variant(S, dtype, budget):
        inp = O.make_inputs(42, S, O.LLAMA2_7B)
        f = {k: (v.to(dtype) if v.dtype == torch.float16 else v) for k, v in inp.items()}
        args = (f["x"], f["residual"], f["weight_qkv"], f["weight_o"], f["k_cache"], f["v_cache"], f["rms_w"], 1e-6,
                f["cos"], f["sin"])
        kw = {"compute_dtype": dtype}
        t_end = time.perf_counter() + budget
        for _ in range(3):
            O.decoder_layer(*args, **kw)
            if time.perf_counter() > t_end:
                break
        times = []
        while len(times) < 30 and (time.perf_counter() < t_end or len(times) < 3):
            t0 = time.perf_counter()
            O.decoder_layer(*args, **kw)
            times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        return {"seq": S, "weights": "fp16 (half matmul)" if dtype == torch.float16 else "fp32 (pre-converted)",
                "us_per_layer": med * 1e6, "tok_s_32_layers": 1.0 / (LAYERS * med), "calls": len(times)}

    v16 = variant(128, torch.float16, budget_s * 0.4)
    v32 = variant(128, torch.float32, budget_s * 0.3)
    vh = variant(S_headline, torch.float32, budget_s * 0.3)
    best = min(v16, v32, key=lambda v: v["us_per_layer"])
    return {"value": best["tok_s_32_layers"], "unit": "tok/s", "cores": torch.get_num_threads(), "kind": "port",
            "us_per_layer": best["us_per_layer"],
            "cpu_model": _cpu_model(), "cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "variants": [v16, v32, vh],
            "sample": "BASELINE config 1: ONE Llama-2-7B layer bs=1 seq=128 (oracle = PyTorch-eager port of the reference layer), "
                      f"3 warm-ups + median of <= 30 calls per variant; `value` = the faster variant ({best['weights']}) as "
                      f"tok/s through 32 such layers; third variant: the headline's seq={S_headline}"}


def _graph_time_us(fn, n_inner, reps, stream):
    """us per call of `fn` (which launches n_inner layer calls), replayed from a HIP graph, HIP events on `stream`."""
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        for _ in range(12):      # (untimed: clock ramp, as in the headline legs)
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_inner)


def other_configs(cfa, dev):
    """The other BASELINE configs on this one GPU (the headline above is config 3): each cycles 32 distinct layer states
    (weights + KV; >= 0.8 GB) so every launch reads HBM; one graph replay = 32 launches, as in the headline run (the replay's
    own launch gap is spread over them alike), HIP events.  frac = algorithmic bytes / time / 8 TB/s."""
    stream = torch.cuda.Stream(dev)
    g = torch.Generator(device=dev).manual_seed(7)

    def rn(*shape):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.1).half()

    out = []

    def record(name, us, S, hq, hkv, residual, note=None):
        b = cfa.algorithmic_bytes(S, HIDDEN, hq, hkv, HEAD_DIM, 1, residual)
        rec = {"name": name, "us_per_call": us, "bytes": b, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "kernel": cfa.last_variant(), "path": cfa.last_path()}
        if note:
            rec["note"] = note
        out.append(rec)

    # ---- config 2: the north-star entry, clusterfusion.llama_decoder_layer ([in,out] weights, GPT-J), S = 1024 --------
    S = 1024
    layers = []
    for _ in range(32):
        ang = torch.rand(HEAD_DIM // 2, generator=g, device=dev) * 6.28
        layers.append((rn(1, 1, HIDDEN), rn(3 * HIDDEN, HIDDEN), rn(HIDDEN, HIDDEN), rn(S, HIDDEN), rn(S, HIDDEN), rn(HIDDEN),
                       ang.cos().repeat_interleave(2).view(1, 128).contiguous(), ang.sin().repeat_interleave(2).view(1, 128).contiguous()))

    def plain():
        for L in layers:
            cfa.llama_decoder_layer(*L)
    cfa.set_weight_relayout(True)
    us = _graph_time_us(plain, len(layers), 20, stream)
    record("config 2: llama_decoder_layer (plain entry, [in,out] weights, GPT-J) S=1024", us, S, HEADS, HEADS, False,
           "default: weights re-laid out once to [out,in] (set_weight_relayout)")
    cfa.set_weight_relayout(False)
    us = _graph_time_us(plain, len(layers), 20, stream)
    record("config 2 (native): same entry, re-layout off -> the [in,out] kernel", us, S, HEADS, HEADS, False)
    cfa.set_weight_relayout(True)
    del layers

    # ---- config 4: Llama-3-8B GQA 32 q / 8 kv heads, S = 8192 --------------------------------------------------------
    def prepared(n, hq, hkv, S, residual=True):
        ls = []
        for _ in range(n):
            ang = torch.rand(HEAD_DIM // 2, generator=g, device=dev) * 6.28
            ls.append(cfa.prepare_decoder_layer(
                rn(1, HIDDEN), rn(1, HIDDEN) if residual else None, rn((hq + 2 * hkv) * HEAD_DIM, HIDDEN),
                rn(HIDDEN, hq * HEAD_DIM), rn(S, hkv * HEAD_DIM), rn(S, hkv * HEAD_DIM), rn(HIDDEN), 1e-6, ang.cos(), ang.sin(),
                n_q_heads=hq, n_kv_heads=hkv, want_kv=True))
        return ls
    # ---- config 1's shape on the GPU (BASELINE.md section 2 prices it: S = 128; the config itself is the CPU baseline) ---------------
    ls = prepared(32, HEADS, HEADS, 128)
    us = _graph_time_us(lambda: [p.run() for p in ls], len(ls), 20, stream)
    record("config 1's shape on the GPU: Llama-2-7B S=128 (sglang entry; the config itself is the CPU baseline)", us, 128, HEADS, HEADS, True)
    del ls
    ls = prepared(32, 32, 8, 8192)
    us = _graph_time_us(lambda: [p.run() for p in ls], len(ls), 20, stream)
    record("config 4: Llama-3-8B GQA 32q/8kv S=8192", us, 8192, 32, 8, True)
    del ls
    # ---- config 5: one rank's shard of head-parallel TP = 8 / 4 / 2 (4 / 8 / 16 heads), S = 4096, before the all-reduce -------
    for tp, hq in ((8, 4), (4, 8), (2, 16)):
        ls = prepared(32, hq, hq, 4096)
        us = _graph_time_us(lambda: [p.run() for p in ls], len(ls), 20, stream)
        record(f"config 5 (per rank): Llama-2-7B TP={tp} shard, {hq} heads, S=4096, local compute before the all-reduce", us, 4096, hq, hq, True)
        del ls
    # ---- config 5 (per rank) with the collective's publish folded into phase 3: the 4-head shard also writes its partial into 8
    #      receive areas (all on this GPU: what that costs the kernel; the xGMI link is not in it), and the gather alone ----------
    from clusterfusion_amd.tp import OneShotReducer
    areas = [torch.zeros(OneShotReducer.area_bytes(8, HIDDEN), dtype=torch.uint8, device=dev) for _ in range(8)]
    red8 = OneShotReducer(0, 8, HIDDEN, areas)
    base4 = prepared(32, 4, 4, 4096)
    ls = [p.with_tp_publish(red8) for p in base4]
    us = _graph_time_us(lambda: [p.run() for p in ls], len(ls), 20, stream)
    record("config 5 (per rank): the TP=8 shard with phase 3 also publishing its partial into 8 receive areas (in-kernel publish of "
           "the one-shot all-reduce; areas on this GPU)", us, 4096, 4, 4, True)
    red1 = OneShotReducer(0, 1, HIDDEN, [torch.zeros(OneShotReducer.area_bytes(1, HIDDEN), dtype=torch.uint8, device=dev)])
    ls1 = [p.with_tp_publish(red1) for p in ls]
    buf = torch.empty(HIDDEN, dtype=torch.float16, device=dev)

    def pub_gather():
        for p in ls1:
            p.run()
            red1.gather(buf)
    us_pg = _graph_time_us(pub_gather, len(ls1), 20, stream)
    out.append({"name": "config 5 (per rank): the same shard + cf_tp_gather behind every layer (world 1: the gather's launch and local poll)",
                "us_per_call": us_pg, "gather_us": us_pg - us, "kernel": cfa.last_variant() + " + k_tp_oneshot_allreduce(gather only)",
                "path": cfa.last_path(), "error_word": red1.error()})
    # ... and with the gather where it belongs in a decoder: inside the fused add + RMSNorm that follows the attention block anyway
    # (cf_rmsnorm_tp_gather: no launch of its own).  EIGHT virtual ranks on this GPU -- every rank's shard kernel publishes into all 8
    # receive areas, every rank's norm polls its 8 slots: the full protocol minus the xGMI hop -- 4 layer states per rank (32 launches per
    # graph replay), against the same shards without TP followed by the plain fused add + RMSNorm
    rw = rn(HIDDEN)
    hres = rn(1, HIDDEN)
    areas8 = [torch.zeros(OneShotReducer.area_bytes(8, HIDDEN), dtype=torch.uint8, device=dev) for _ in range(8)]
    reds8 = [OneShotReducer(r, 8, HIDDEN, areas8) for r in range(8)]
    vbase = [base4[4 * r: 4 * r + 4] for r in range(8)]                      # rank r's 4 layer states
    vpub = [[p.with_tp_publish(reds8[r]) for p in vbase[r]] for r in range(8)]
    houts = [torch.empty(1, HIDDEN, dtype=torch.float16, device=dev) for _ in range(8)]

    def v_pub_norm_gather():
        for l in range(4):
            for r in range(8):
                vpub[r][l].run()
            for r in range(8):
                reds8[r].gather_rmsnorm(rw, 1e-6, residual=hres, out=houts[r])

    def v_plain_norm():
        for l in range(4):
            for r in range(8):
                vbase[r][l].run()
            for r in range(8):
                cfa.rmsnorm(vbase[r][l].outputs[0], rw, 1e-6, residual=hres, out=houts[r])
    us_png = _graph_time_us(v_pub_norm_gather, 32, 20, stream)
    us_pn = _graph_time_us(v_plain_norm, 32, 20, stream)
    out.append({"name": "config 5 (per rank): TP=8 shard with the in-kernel publish + the gather folded into the next fused add + RMSNorm "
                        "(cf_rmsnorm_tp_gather, 8 virtual ranks on this GPU: the whole protocol minus the xGMI hop) -- one layer's attention "
                        "block AND the norm behind it",
                "us_per_call": us_png, "same_without_tp_us": us_pn, "collective_cost_us": us_png - us_pn,
                "note": "same_without_tp_us = the shard kernel without publish + clusterfusion.rmsnorm(residual=...): what the two launches cost "
                        "when no collective is involved; the difference is what the library's all-reduce adds per layer on this GPU",
                "kernel": "k_fused_decode_s<4> + k_rmsnorm_tp_gather_mw", "path": cfa.last_path(), "error_word": max(r.error() for r in reds8)})
    del vbase, vpub, reds8, areas8
    del ls, ls1, base4
    # ---- configs 4 and 5 composed: one rank's shard of head-parallel TP = 2 / 4 / 8 of Llama-3-8B (16q/4kv, 8q/2kv, 4q/1kv), S = 8192 ----
    for tp, hq, hkv in ((2, 16, 4), (4, 8, 2), (8, 4, 1)):
        ls = prepared(32, hq, hkv, 8192)
        us = _graph_time_us(lambda: [p.run() for p in ls], len(ls), 20, stream)
        record(f"configs 4 + 5 (per rank): Llama-3-8B GQA TP={tp} shard, {hq}q/{hkv}kv heads, S=8192, local compute before the all-reduce", us, 8192, hq, hkv, True)
        del ls
    # ---- the reference's batched entry with 2 / 4 / 8 / 16 / 32 sequences (Llama-2-7B, paged KV, page size 1, S = 1024 each) ------
    S, NL = 1024, 32
    wq = [rn(3 * HIDDEN, HIDDEN) for _ in range(NL)]
    wo = [rn(HIDDEN, HIDDEN) for _ in range(NL)]
    rms = [rn(HIDDEN) for _ in range(NL)]
    for bs in (2, 4, 8, 16, 32):
        n_slots = bs * (S + 1)
        kcs = [rn(n_slots, HIDDEN) for _ in range(NL)]
        vcs = [rn(n_slots, HIDDEN) for _ in range(NL)]
        kptrs = torch.tensor([t.data_ptr() for t in kcs], dtype=torch.uint64, device=dev)
        vptrs = torch.tensor([t.data_ptr() for t in vcs], dtype=torch.uint64, device=dev)
        perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(bs)).to(torch.int32).to(dev)
        indptr = (torch.arange(bs + 1, dtype=torch.int32) * (S + 1)).to(dev)
        positions = torch.full((bs,), S, dtype=torch.int64, device=dev)
        cos_sin = (torch.rand(S + 1, 128, generator=g, device=dev) * 2 - 1).float()
        x, r = rn(bs, HIDDEN), rn(bs, HIDDEN)
        o, ro = torch.empty_like(x), torch.empty_like(x)

        def batched():
            for l in range(NL):
                cfa.llama_decoder_layer_batch_decode_sglang(o, ro, x, r, wq[l], wo[l], indptr, perm, kptrs, vptrs, l, rms[l], 1e-6,
                                                            positions, cos_sin)
        us = _graph_time_us(batched, NL, 20, stream)
        b = 2 * HIDDEN * 3 * HIDDEN + 2 * HIDDEN * HIDDEN + bs * 4 * S * HIDDEN      # weights once + every row's K/V
        out.append({"name": f"llama_decoder_layer_batch_decode_sglang, {bs} sequences x S=1024 (paged, page size 1)", "us_per_call": us,
                    "bytes": b, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel": cfa.last_variant(), "path": cfa.last_path()})
        del kcs, vcs
    # ---- configs 4 and f2 composed: 2 / 4 sequences of the Llama-3-8B geometry (32q/8kv), S = 8192 each, one persistent launch ----
    S, NL = 8192, 32
    wq = [rn((32 + 16) * HEAD_DIM, HIDDEN) for _ in range(NL)]
    wo = [rn(HIDDEN, HIDDEN) for _ in range(NL)]
    rms = [rn(HIDDEN) for _ in range(NL)]
    for bs in (2, 4):
        n_slots = bs * (S + 1)
        kcs = [rn(n_slots, 8 * HEAD_DIM) for _ in range(NL)]
        vcs = [rn(n_slots, 8 * HEAD_DIM) for _ in range(NL)]
        perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(100 + bs)).to(torch.int32).to(dev)
        indptr = (torch.arange(bs + 1, dtype=torch.int32) * (S + 1)).to(dev)
        positions = torch.full((bs,), S, dtype=torch.int64, device=dev)
        lens = torch.full((bs,), S, dtype=torch.int32, device=dev)
        cos_sin = (torch.rand(S + 1, 128, generator=g, device=dev) * 2 - 1).float()
        x, r = rn(bs, HIDDEN), rn(bs, HIDDEN)
        ls = [cfa.prepare_decoder_layer(x, r, wq[l], wo[l], kcs[l], vcs[l], rms[l], 1e-6, cos_sin, cos_sin.view(-1)[64:], n_q_heads=32,
                                        n_kv_heads=8, kv_indptr=indptr, kv_indices=perm, kv_seq_lens=lens, page_size=1, positions=positions,
                                        rope_row_stride=128, write_kv_to_cache=True, max_seq_len=S) for l in range(NL)]
        us = _graph_time_us(lambda: [p.run() for p in ls], NL, 20, stream)
        b = 2 * HIDDEN * 48 * HEAD_DIM + 2 * HIDDEN * HIDDEN + bs * 4 * S * 8 * HEAD_DIM      # weights once + every row's K/V
        out.append({"name": f"configs 4 + f2: Llama-3-8B GQA 32q/8kv, {bs} sequences x S=8192 (paged, page size 1), one launch", "us_per_call": us,
                    "bytes": b, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel": cfa.last_variant(), "path": cfa.last_path()})
        del kcs, vcs, ls
    del wq, wo, rms
    # ---- the reference's second model family: deepseek_decoder_layer (DeepSeek-V2-Lite MLA block), S = 4096, 27 layers ----
    import math
    H, N, R, L, D, S = 16, 128, 64, 512, 2048, 4096

    def rs(scale, *shape):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * scale).half()
    pos = float(S - 1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, R, 2, dtype=torch.float64) / R))
    ang = torch.cat([pos * inv, pos * inv])
    cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
    mla = [[rs(1.0, 1, D), rs(1 / math.sqrt(D), D, H * N), rs(1 / math.sqrt(D), D, H * R),
            rs(3.0 / math.sqrt(N) * math.sqrt((N + R) / L), N, H * L), rs(1 / math.sqrt(D), D, L), rs(1 / math.sqrt(D), D, R),
            rs(1 / math.sqrt(L), L, H * N), rs(1 / math.sqrt(H * N), H * N, D), rs(1.0, S, L + R),
            (1.0 + rs(0.1, D).float()).half(), (1.0 + rs(0.1, L).float()).half(), cos, sin] for _ in range(27)]
    us = _graph_time_us(lambda: [cfa.deepseek_decoder_layer(*m) for m in mla], len(mla), 20, stream)
    b = cfa.deepseek_algorithmic_bytes(S, False)
    out.append({"name": "deepseek_decoder_layer (DeepSeek-V2-Lite MLA attention block), S=4096, 27 distinct layers", "us_per_call": us,
                "bytes": b, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel": "k_mla_fused" if cfa.last_path() == "fused" else "3 launches",
                "path": cfa.last_path(), "note": "latency chain of five hand-offs (DESIGN 3.4); parity unpinned (no reference test exists)"})
    del mla
    torch.cuda.empty_cache()
    # ---- the op in its real place -- a whole Llama-2-7B-shaped decoder, greedy decode, one graph per token --
    from clusterfusion_amd.harness import DecodeModel
    S0, steps = 4000, 48      # (stays below 4097 cached tokens: the two-tile arm, as the headline)
    m = DecodeModel(start_pos=S0, max_seq=S0 + 3 * steps + 64)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(2):
            m.step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            m.step()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gr.replay()                      # ONE captured graph while the sequence grows past 4096 cached tokens
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        arm = cfa.last_arm()
    out.append({"name": "whole-model greedy decode, Llama-2-7B shapes, ~4000 cached tokens (attention block = the fused op, "
                        "norms = clusterfusion.rmsnorm; SwiGLU FFN / LM head / argmax = torch: outside the reference's op too)",
                "tok_s": 1e3 / ms, "ms_per_token": ms, "kernel": cfa.last_variant(), "path": cfa.last_path(), "arm_at_end": arm,
                "note": f"one hipGraph captured at S={S0 + 2} and replayed {steps + 3} times while the sequence grows (S={int(m.pos.item())} at "
                        "the end): the kernel reads the length on the device"})
    del m, gr
    torch.cuda.empty_cache()
    return out


def eager_entries(dev):
    """The call the reference's caller actually makes: `clusterfusion.<entry>(...)` EAGERLY, once per layer per token
    (chat/llama/model.py:358-367), through the drop-in package name -- no PreparedLayer, no graph, outputs allocated by the op,
    fresh `cache[:, :start_pos]` views and `rotary[start_pos:start_pos+1]` slices made by the caller before every call.  Per
    entry, over 32 distinct layer states (every launch reads HBM):
      graph_us_per_call   the same 32 calls captured once and replayed (what every other figure of this line is)
      eager_us_per_call   wall clock of the eager loop incl. the final synchronize / calls
      host_us_per_call    wall clock until the loop has ISSUED its last call (before the synchronize) / calls: caller's views + op
      op_host_us_per_call the same with the views made beforehand: the op's own host time per call
    (median of 7 rounds of 4 x 32 calls each).  The KV-cache write-back that follows the call in model.py:370-371 is the
    caller's own two copy kernels and is not in any of them."""
    import clusterfusion                     # the drop-in name (clusterfusion/__init__.py), as the reference's caller imports it
    import clusterfusion_amd as cfa
    stream = torch.cuda.Stream(dev)
    g = torch.Generator(device=dev).manual_seed(11)
    NL, REP, ROUNDS = 32, 4, 7

    def rn(*shape):
        return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * 0.1).half()

    def measure(name, make_call, premade_call, S, bytes_call):
        """make_call(l) -> closure args made the caller's way + the call; premade_call(l): arguments made beforehand"""
        with torch.cuda.stream(stream):
            def loop():
                for l in range(NL):
                    make_call(l)

            def loop_pre():
                for l in range(NL):
                    premade_call(l)
            loop()
            torch.cuda.synchronize()
            graph_us = _graph_time_us(loop, NL, 20, stream)
            for _ in range(3):
                loop()
            torch.cuda.synchronize()
            eager, host, op_host = [], [], []
            for _ in range(ROUNDS):
                t0 = time.perf_counter()
                for _ in range(REP):
                    loop()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                eager.append((t2 - t0) / (REP * NL) * 1e6)
                host.append((t1 - t0) / (REP * NL) * 1e6)
                t0 = time.perf_counter()
                for _ in range(REP):
                    loop_pre()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                op_host.append((t1 - t0) / (REP * NL) * 1e6)
        med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
        e, h, oh = med(eager), med(host), med(op_host)
        return {"name": name, "graph_us_per_call": graph_us, "eager_us_per_call": e, "eager_over_graph": e / graph_us,
                "host_us_per_call": h, "op_host_us_per_call": oh, "bound": "gpu" if h < 0.9 * e else "host",
                "binding": cfa.host_binding(), "bytes": bytes_call, "frac_eager": bytes_call / (e * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "kernel": cfa.last_variant(), "path": cfa.last_path()}

    out = []
    # ---- llama_decoder_layer: the chat demo's call, [in,out] weights, GPT-J cos/sin rows, S = 1024 (config 2) ---------------
    S, MAXS = 1024, 1024 + 64
    ang = torch.rand(MAXS, HEAD_DIM // 2, generator=g, device=dev) * 6.28
    rot_cos, rot_sin = ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous()
    Ls = [dict(wq=rn(3 * HIDDEN, HIDDEN), wo=rn(HIDDEN, HIDDEN), ck=rn(1, MAXS, HEADS, HEAD_DIM), cv=rn(1, MAXS, HEADS, HEAD_DIM),
               rms=rn(HIDDEN)) for _ in range(NL)]
    x = rn(1, 1, HIDDEN)

    def plain_call(l):                       # model.py:353-367, line for line in meaning
        L = Ls[l]
        kk = L["ck"][:1, :S].view(-1, HEADS * HEAD_DIM)
        vv = L["cv"][:1, :S].view(-1, HEADS * HEAD_DIM)
        o, xk, xv = clusterfusion.llama_decoder_layer(x, L["wq"], L["wo"], kk, vv, L["rms"], rot_cos[S:S + 1].to(device=x.device),
                                                      rot_sin[S:S + 1].to(device=x.device))
        return o.view(1, 1, HIDDEN)
    pre = [(L["ck"][:1, :S].view(-1, HIDDEN), L["cv"][:1, :S].view(-1, HIDDEN)) for L in Ls]
    c1, s1 = rot_cos[S:S + 1], rot_sin[S:S + 1]

    def plain_pre(l):
        L = Ls[l]
        return clusterfusion.llama_decoder_layer(x, L["wq"], L["wo"], pre[l][0], pre[l][1], L["rms"], c1, s1)
    cfa.set_weight_relayout(True)
    out.append(measure("eager drop-in call: clusterfusion.llama_decoder_layer as chat/llama/model.py:358-367 issues it, S=1024",
                       plain_call, plain_pre, S, cfa.algorithmic_bytes(S, HIDDEN, HEADS, HEADS, HEAD_DIM, 1, False)))
    cfa.release_weight_relayout()
    del Ls, pre
    # ---- llama_decoder_layer_sglang: tests/test_llama.py:145-156, [out,in] weights, residual in place, S = 4096 ----------------
    S, MAXS = 4096, 4096 + 64
    Ls = [dict(wq=rn(3 * HIDDEN, HIDDEN), wo=rn(HIDDEN, HIDDEN), ck=rn(1, MAXS, HEADS, HEAD_DIM), cv=rn(1, MAXS, HEADS, HEAD_DIM),
               rms=rn(HIDDEN)) for _ in range(NL)]
    ang = torch.rand(1, HEAD_DIM // 2, generator=g, device=dev) * 6.28
    cs, sn = torch.cat([ang.cos(), ang.cos()], 1).contiguous(), torch.cat([ang.sin(), ang.sin()], 1).contiguous()
    x2, res = rn(1, HIDDEN), rn(1, HIDDEN)

    def sg_call(l):
        L = Ls[l]
        kk = L["ck"][:1, :S].view(-1, HEADS * HEAD_DIM)
        vv = L["cv"][:1, :S].view(-1, HEADS * HEAD_DIM)
        return clusterfusion.llama_decoder_layer_sglang(x2, res, L["wq"], L["wo"], kk, vv, L["rms"], 1e-6, cs, sn)
    pre = [(L["ck"][:1, :S].view(-1, HIDDEN), L["cv"][:1, :S].view(-1, HIDDEN)) for L in Ls]

    def sg_pre(l):
        L = Ls[l]
        return clusterfusion.llama_decoder_layer_sglang(x2, res, L["wq"], L["wo"], pre[l][0], pre[l][1], L["rms"], 1e-6, cs, sn)
    out.append(measure("eager drop-in call: clusterfusion.llama_decoder_layer_sglang as tests/test_llama.py:145-156 issues it, S=4096",
                       sg_call, sg_pre, S, cfa.algorithmic_bytes(S, HIDDEN, HEADS, HEADS, HEAD_DIM, 1, True)))
    del pre
    # ---- llama_decoder_layer_batch_decode_sglang: 1 and 4 sequences of S = 1024, page size 1 (the caller owns every buffer) ----
    S = 1024
    for bs in (1, 4):
        n_slots = bs * (S + 1)
        kptrs = torch.tensor([L["ck"].data_ptr() for L in Ls], dtype=torch.uint64, device=dev)
        vptrs = torch.tensor([L["cv"].data_ptr() for L in Ls], dtype=torch.uint64, device=dev)
        perm = torch.randperm(n_slots, generator=torch.Generator().manual_seed(bs)).to(torch.int32).to(dev)
        indptr = (torch.arange(bs + 1, dtype=torch.int32) * (S + 1)).to(dev)
        positions = torch.full((bs,), S, dtype=torch.int64, device=dev)
        cos_sin = (torch.rand(S + 1, 128, generator=g, device=dev) * 2 - 1).float()
        xb, rb = rn(bs, HIDDEN), rn(bs, HIDDEN)
        ob, rob = torch.empty_like(xb), torch.empty_like(xb)

        def b_call(l):
            L = Ls[l]
            clusterfusion.llama_decoder_layer_batch_decode_sglang(ob, rob, xb, rb, L["wq"], L["wo"], indptr, perm, kptrs, vptrs, l, L["rms"],
                                                                  1e-6, positions, cos_sin)
        b = 2 * HIDDEN * 3 * HIDDEN + 2 * HIDDEN * HIDDEN + bs * 4 * S * HIDDEN
        out.append(measure(f"eager drop-in call: clusterfusion.llama_decoder_layer_batch_decode_sglang, {bs} sequence(s) x S=1024, page size 1",
                           b_call, b_call, S, b))
    del Ls
    # ---- config 4 through the same entry name: weight_qkv [6144, 4096] = 32 q / 8 kv heads (an extension of the entry), S = 8192 -------------
    S, MAXS, KVD = 8192, 8192 + 64, 8 * HEAD_DIM
    Ls = [dict(wq=rn((HEADS + 16) * HEAD_DIM, HIDDEN), wo=rn(HIDDEN, HIDDEN), ck=rn(1, MAXS, 8, HEAD_DIM), cv=rn(1, MAXS, 8, HEAD_DIM),
               rms=rn(HIDDEN)) for _ in range(NL)]

    def gq_call(l):
        L = Ls[l]
        kk = L["ck"][:1, :S].view(-1, KVD)
        vv = L["cv"][:1, :S].view(-1, KVD)
        return clusterfusion.llama_decoder_layer_sglang(x2, res, L["wq"], L["wo"], kk, vv, L["rms"], 1e-6, cs, sn)
    pre = [(L["ck"][:1, :S].view(-1, KVD), L["cv"][:1, :S].view(-1, KVD)) for L in Ls]

    def gq_pre(l):
        L = Ls[l]
        return clusterfusion.llama_decoder_layer_sglang(x2, res, L["wq"], L["wo"], pre[l][0], pre[l][1], L["rms"], 1e-6, cs, sn)
    out.append(measure("eager drop-in call: clusterfusion.llama_decoder_layer_sglang with Llama-3-8B weights (32q/8kv: config 4), S=8192",
                       gq_call, gq_pre, S, cfa.algorithmic_bytes(S, HIDDEN, HEADS, 8, HEAD_DIM, 1, True)))
    del Ls, pre
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(a):
    """`python bench.py --gpus N` as the driver calls it (no WORLD_SIZE in the environment): re-launch this script as N ranks,
    one per GPU, through torch.distributed.run on 127.0.0.1; rank 0's JSON line is the last line of the inherited stdout."""
    import subprocess
    if not a.dry_launch:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible to this process "
                             "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL and the peer-mapped receive areas need here
    env.setdefault("OMP_NUM_THREADS", "8")
    argv = [x for x in sys.argv[1:] if x != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_launch(a, world, rank):
    """The launch path without a GPU (`-m "not gpu"` test): rendezvous over gloo, one all-reduce, one JSON line from rank 0."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise SystemExit("bench.py --dry-launch: WORLD_SIZE > 1 without MASTER_PORT (launch through torch.distributed.run or --gpus N)")
        os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "gpus_arg": a.gpus, "all_reduce_sum": t.item(),
                          "expected": world * (world + 1) / 2}), flush=True)


def tp_parity(layers_l0, full_l0, outs_buf, reduce_fn, use_dist, rank, world, dev, precomputed=None):
    """One seeded full layer through shard + collective on every rank against the SAME layer through the 1-GPU kernel on rank 0
    (no oracle here: the unsharded HIP kernel is the reference; it is itself held to the oracle by tests/test_parity_gpu.py).
    `layers_l0` = this process's shard(s) of layer 0 (several when it plays more than one rank: their fp16 partials are summed
    in fp32 and rounded once before the collective)."""
    if precomputed is not None:      # (the caller ran shard + collective itself: the in-kernel publish leg)
        red = precomputed
    else:
        parts = [p.run()[0] for p in layers_l0]
        if len(parts) == 1:
            red = parts[0]
        else:
            red = outs_buf
            red.copy_(torch.stack([q.float() for q in parts]).sum(0).half())
        reduce_fn(red)
    torch.cuda.synchronize()
    mine = red.clone()
    ref = torch.empty_like(mine)
    if rank == 0:
        ref.copy_(full_l0.run()[0])
        torch.cuda.synchronize()
    if use_dist:
        dist.broadcast(ref, 0)
    err = (mine.float() - ref.float()).abs().max()
    same = torch.ones(1, device=dev)
    if use_dist:
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = torch.tensor([float(all(torch.equal(gathered[0], t) for t in gathered))], device=dev)
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
    err, same = err.item(), bool(same.item())
    return {"max_abs_err_vs_1gpu_kernel": err, "tol": 2e-3, "identical_bits_on_every_rank": same,
            "ok": bool(err <= 2e-3 and same), "max_abs_ref": ref.float().abs().max().item()}


def main():
    a = parse()
    if (a.gpus > 1 or a.spawn) and "WORLD_SIZE" not in os.environ:
        self_spawn(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_launch:
        return dry_launch(a, world, rank)
    # CF_BENCH_TP=N (debug): run the per-rank workload of an N-way head-parallel shard on however many
    # ranks were launched (lets a 1-GPU box exercise the TP code path incl. the RCCL all-reduce)
    tp = int(os.environ.get("CF_BENCH_TP", str(world)))
    force_dist = os.environ.get("CF_BENCH_FORCE_DIST", "0") == "1"
    if world != a.gpus:
        a.gpus = world          # (launched through torch.distributed.run with another rank count: the launcher wins)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # (only a lone CF_BENCH_FORCE_DIST process gets here without a launcher's port)
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    import clusterfusion_amd as cfa
    if a.only_eager:
        print(json.dumps(eager_entries(dev), indent=1))
        return
    if a.kv_splits:
        cfa.set_tuning(a.kv_splits)
    if a.debug_flags:
        from clusterfusion_amd import _lib
        _lib.load().cf_debug_set_flags(a.debug_flags)
    cfa.set_path(a.path)
    S = a.seq
    if tp % world or HEADS % tp:
        raise SystemExit(f"bench.py: a {tp}-way head shard cannot be dealt to {world} rank(s)")
    virt = list(range(rank, tp, world))      # the shard ranks this process plays (one, unless CF_BENCH_TP > WORLD_SIZE)
    layers, full, extra = build_layers(cfa, dev, tp, rank, a.layers, S, a.page_size, keep_full=1 if tp > 1 else 0, extra_ranks=virt[1:])
    outs = [p.outputs[0] for p in layers]

    # the collective of the head-parallel path: RCCL's all-reduce is the measured contract (north_star); the library's one-shot
    # all-reduce over peer-mapped buffers (clusterfusion_amd.tp.OneShotReducer) runs as a second leg
    def make_step(reduce_fn):
        def step():
            for p, o in zip(layers, outs):
                p.run()
                if reduce_fn is not None:
                    reduce_fn(o)
        return step

    def rccl(o):
        dist.all_reduce(o)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()        # synthetic state was drawn on the default stream
    stream = torch.cuda.Stream(dev)

    class LegUnhealthy(RuntimeError):
        pass
    host_issue_s = [0.0]

    def timed_leg(reduce_fn, step_fn=None, health=None):
        """warm-up, then EXACTLY a.steps steps between barrier + synchronize on both sides; max over ranks.  `health`: checked on
        every rank after the first step (a collective that starts timing out -- 2 s per call -- must not eat the run)."""
        step = step_fn if step_fn is not None else make_step(reduce_fn)
        graph = None
        step()                      # first call: lazy init (workspace, RCCL channels)
        torch.cuda.synchronize()
        if health is not None:
            ok = torch.tensor([float(bool(health()))], device=dev)
            if use_dist:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not ok.item():
                raise LegUnhealthy("the collective reported an error after the first step")
        if not a.no_graph:
            try:
                if use_dist:
                    # RCCL's watchdog thread polls the events of the eager collectives above with hipEventQuery every ~100 ms.  Under
                    # the default ("global") capture mode a query from ANY thread while this one captures invalidates the capture:
                    # that was the 1-in-10..40 abort of this entry (profiles/r06_spawn_soak.md).  Let the watchdog retire what is
                    # already complete, then capture in thread-local mode (other threads' calls are not this capture's business).
                    time.sleep(0.3)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local" if use_dist else "global"):
                    step()
                graph = g
            except Exception as e:   # noqa: BLE001 -- capture is an optimisation of the host side only
                print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        run = graph.replay if graph is not None else step
        # Clock ramp: the first ~30 ms of back-to-back launches after an idle period run 0.4-0.8 % slower than the steady state a
        # decode loop lives in (same-box alternation, --steps 20: 35.50-35.65 us per layer behind 5 warm-up steps, 35.34-35.40
        # behind 40).  PREWARM untimed steps bring the part to its steady clocks; then the W warm-up steps the caller asked for.
        for _ in range(PREWARM):
            run()
        for _ in range(a.warmup):
            run()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)          # HIP events on the stream the kernels are launched on
        for _ in range(a.steps):
            run()
        ev1.record(stream)
        host_issue_s[0] = time.perf_counter() - t0      # the host's share: all steps ISSUED (graph launches or eager calls), nothing waited for
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        # the collective alone (every rank takes part): eager back-to-back calls on one layer's output, HIP events on the stream
        coll_us = None
        if reduce_fn is not None:
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_coll = 200
            barrier()
            c0.record(stream)
            for _ in range(n_coll):
                reduce_fn(outs[0])
            c1.record(stream)
            barrier()
            coll_us = c0.elapsed_time(c1) * 1e3 / n_coll
        if use_dist:
            t = torch.tensor([dt, coll_us or 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, coll_us = t[0].item(), (t[1].item() if reduce_fn is not None else None)
        return dt, ev_ms, coll_us, graph is not None

    parity, oneshot_rec, inkernel_rec = None, None, None
    dt = ev_ms = coll_us = graphed = kernel_variant = layer_path = None
    host_issue_us = 0.0
    stage_us, rank_kernel_us = [0.0] * 4, [0.0]

    def build_rec():
        ms_per_step = dt / a.steps * 1e3
        us_layer = ms_per_step * 1e3 / a.layers
        hq = HEADS // tp
        bytes_layer = cfa.algorithmic_bytes(S, HIDDEN, hq, hq, HEAD_DIM, 1, True)
        path = layer_path      # (read on the launching thread: the library keeps it per thread)
        if path == "fused":
            # ONE persistent kernel per layer: its algorithmic bytes are the layer's
            # duration: HIP-event pair around the timed region on the launch stream / number of launches
            # (the launches are back to back, so this includes the ~0.3 us inter-launch gap; events
            # recorded BETWEEN launches would add their own ~4 us of command-processor gap each)
            kern_name, kern_bytes = kernel_variant + " (whole layer, one persistent launch)", bytes_layer
            kern_us = ev_ms * 1e3 / (a.steps * a.layers)
            if use_dist:   # the timed region also holds the all-reduce: take the kernel alone (library events), slowest rank
                kern_us = max(rank_kernel_us)
        else:
            # dominant kernel = stage 0 (RMSNorm + QKV projection): Wqkv shard + x, residual, rms_w, raw q|k|v out
            kern_name = "k_qkv_rows (RMSNorm + QKV GEMV)"
            kern_bytes = 2 * HIDDEN * 3 * hq * HEAD_DIM + 3 * 2 * HIDDEN + 4 * 3 * hq * HEAD_DIM
            kern_us = max(rank_kernel_us)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(kern_name.split(" ")[0].split("<")[0] + "_bytes_per_launch") if tp == 1 else None
                traffic_source = "recorded counter figure, not measured in this run: " + tj.get("_source", tpath) if traffic else None
            except Exception:   # noqa: BLE001
                traffic = None
        roof = {"bound": "hbm", "kernel": kern_name,
                "achieved": kern_bytes / (kern_us * 1e-6) / 1e9 if kern_us > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (kern_bytes / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if kern_us > 0 else None,
                "traffic": traffic, "traffic_source": traffic_source, "bytes_per_launch": kern_bytes, "us_per_launch": kern_us,
                "timing": ("hipEvents recorded by the library around the kernel, eager launches (the timed region also holds the all-reduce); "
                           "slowest rank, every rank's figure in us_per_launch_by_rank"
                           if path == "fused" and use_dist else
                           "HIP events around the timed region on the launch stream / launches" if path == "fused" else
                           "hipEvents recorded by the library on its launch stream around each kernel, eager launches"),
                "us_per_launch_events_between": stage_us[0],
                "stage_us": {"qkv_or_fused": stage_us[0], "attention": stage_us[1], "oproj": stage_us[2],
                             "reduce": stage_us[3]},
                "layer": {"bytes": bytes_layer, "us": us_layer,
                          "achieved": bytes_layer / (us_layer * 1e-6) / 1e9,
                          "frac": bytes_layer / (us_layer * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "timing": "wall clock of the timed region / (steps x layers)"}}
        if use_dist:
            roof["us_per_launch_by_rank"] = [round(x, 3) for x in rank_kernel_us]
        rec = {
            "metric": "decode tok/s through the fused attention-block op of 32 layers (us/decoder-layer alongside), "
                      "Llama-2-7B bs=1 seq=4096",
            "value": 1e3 / (ms_per_step * LAYERS / a.layers),
            "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_steps": PREWARM, "ms_per_step": ms_per_step,
            "us_per_layer": us_layer, "higher_is_better": True,
            "scaling": "strong",     # the model is fixed: N GPUs share ONE sequence's layer (head-parallel), N = 1 is that curve's first point
            "vs_baseline": None, "dtype": "f16 storage, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"Llama-2-7B fused attention-block decode, bs=1 seq={S}, paged KV page_size="
                                   f"{a.page_size}, {a.layers} distinct layers per step (BASELINE configs[2]"
                                   + (f" sharded head-parallel TP={tp}, one all-reduce per layer = configs[4])"
                                      if tp > 1 else ")"),
                       "parallelism": f"tp{tp}", "collective": "RCCL all_reduce" if use_dist else None,
                       "collective_us_alone": None if coll_us is None else round(coll_us, 2),
                       "launch": "hipGraph replay" if graphed else "eager",
                       # host time to ISSUE one step (no waiting): next to ms_per_step it says whether a curve is host- or GPU-bound
                       "host_us_per_step": round(host_issue_us, 1),
                       "kv_splits": a.kv_splits or "auto", "path": path},
            "roofline": roof,
        }
        if parity is not None:
            rec["tp_parity"] = parity
        if oneshot_rec is not None:
            rec["oneshot"] = oneshot_rec
        if inkernel_rec is not None:
            rec["inkernel_publish"] = inkernel_rec
        return rec

    # The optional legs (one-shot all-reduce, in-kernel publish) hold collectives of their own: a rank that drops out of one of
    # them (an IPC mapping that fails on one GPU only, a launch error) would leave the others waiting in RCCL for its watchdog's
    # minutes.  They must never cost the headline line: past --leg-timeout seconds rank 0 prints what it has -- the RCCL leg is
    # complete by then -- and every rank leaves.
    emit_lock, emitted, watchdog = threading.Lock(), [False], None

    def print_rec(rec):
        # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the LAST line
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        print(json.dumps(rec), flush=True)

    def bail():
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            if rank == 0:
                why = f"gave up after {a.leg_timeout:.0f} s (a rank dropped out of the leg or a collective hung); the lines above it are complete"
                for r in (oneshot_rec, inkernel_rec):
                    if r is not None and "status" not in r:
                        r["status"] = why
                try:
                    print_rec(build_rec())
                finally:
                    os._exit(0)
            os._exit(0)

    with torch.cuda.stream(stream):
        if tp > 1:
            l0 = [layers[0]] + extra[0]
            buf = torch.empty_like(outs[0])
            parity = {"collective": "RCCL all_reduce",
                      **tp_parity(l0, full[0], buf, rccl if use_dist else (lambda o: o), use_dist, rank, world, dev)}
        dt, ev_ms, coll_us, graphed = timed_leg(rccl if use_dist else None)
        host_issue_us = host_issue_s[0] / a.steps * 1e6      # (of the headline leg: the optional legs below overwrite the cell)

        # per-kernel durations: HIP events recorded by the library on ITS launch stream around each
        # kernel, eager launches, same workload (a separate pass so the events do not sit in `dt`); every rank, slowest reported
        cfa.profile_enable(True)
        for _ in range(max(2, min(a.steps, 20))):
            for p in layers:
                p.run()
        torch.cuda.synchronize()
        stage_ms, ncalls = cfa.profile_read(reset=True)
        cfa.profile_enable(False)
        kernel_variant, layer_path = cfa.last_variant(), cfa.last_path()
        stage_us = [m * 1e3 / max(ncalls, 1) for m in stage_ms]
        rank_kernel_us = [stage_us[0]]
        if use_dist:
            t = torch.tensor([stage_us[0]], dtype=torch.float64, device=dev)
            allk = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allk, t)
            rank_kernel_us = [x.item() for x in allk]

        # ---- second leg: the library's one-shot all-reduce (unmeasured over xGMI until a multi-GPU box runs this) -----------
        if use_dist and not a.no_oneshot:
            watchdog = threading.Timer(a.leg_timeout, bail)
            watchdog.daemon = True
            watchdog.start()
            oneshot_rec = {"collective": "one-shot (cf_tp_oneshot_allreduce): every rank writes its 8 KB partial into its slot of "
                                         "every peer's receive area, polls its own, sums in rank order"}
            try:
                if os.environ.get("CF_BENCH_FAULT") == "hang_leg":      # (test hook: what a hung collective looks like to the watchdog)
                    time.sleep(1e6)
                from clusterfusion_amd.tp import OneShotReducer
                red = OneShotReducer.create(None, HIDDEN, dev)
                probe = torch.full((HIDDEN,), float(rank + 1), dtype=torch.float16, device=dev)
                try:                     # self-check first: a link that does not carry the protocol must not eat the run
                    red(probe)
                    torch.cuda.synchronize()
                    probe_ok = red.error() == 0 and bool((probe == world * (world + 1) / 2).all())
                except Exception as e:   # noqa: BLE001 -- every rank still takes part in the agreement below
                    print(f"[bench] rank {rank}: one-shot self-check raised {type(e).__name__}: {e}", file=sys.stderr)
                    probe_ok = False
                good = torch.tensor([float(probe_ok)], device=dev)
                dist.all_reduce(good, op=dist.ReduceOp.MIN)
                if not good.item():
                    oneshot_rec["status"] = f"self-check failed on some rank (this rank: error word {red.error()}); leg skipped"
                else:
                    def oneshot(o):
                        red(o.view(-1))
                    if tp > 1:
                        oneshot_rec["tp_parity"] = tp_parity([layers[0]] + extra[0], full[0], torch.empty_like(outs[0]), oneshot,
                                                             use_dist, rank, world, dev)
                    dt1, _, coll1, _ = timed_leg(oneshot, health=lambda: red.error() == 0)
                    codes = torch.tensor([float(red.error())], device=dev)
                    dist.all_reduce(codes, op=dist.ReduceOp.MAX)
                    oneshot_rec.update({"status": "ok" if codes.item() == 0 else f"error word {int(codes.item())} after the timed leg",
                                        "ms_per_step": dt1 / a.steps * 1e3, "us_per_layer": dt1 / a.steps * 1e6 / a.layers,
                                        "tok_s": a.steps / dt1 * a.layers / LAYERS, "collective_us_alone": round(coll1, 2)})
                    # ---- third leg: the publish folded into phase 3 of the shard kernels, the gather as its own (local) launch ----
                    if codes.item() == 0:
                        inkernel_rec = {"collective": "publish in phase 3 of the layer kernel (cf_layer_args.tp_areas) + cf_tp_gather: "
                                                      "no publish launch, no re-read of the partial"}
                        pub_layers = [p.with_tp_publish(red) for p in layers]

                        def step_pub():
                            for p, o in zip(pub_layers, outs):
                                p.run()
                                red.gather(o.view(-1))
                        if tp > 1 and len(virt) == 1:
                            pub_layers[0].run()
                            red.gather(outs[0].view(-1))
                            torch.cuda.synchronize()
                            inkernel_rec["tp_parity"] = tp_parity([layers[0]], full[0], None, lambda o: None, use_dist, rank, world, dev,
                                                                  precomputed=outs[0])
                        dt2, _, _, _ = timed_leg(None, step_fn=step_pub, health=lambda: red.error() == 0)
                        codes = torch.tensor([float(red.error())], device=dev)
                        dist.all_reduce(codes, op=dist.ReduceOp.MAX)
                        inkernel_rec.update({"status": "ok" if codes.item() == 0 else f"error word {int(codes.item())} after the timed leg",
                                             "ms_per_step": dt2 / a.steps * 1e3, "us_per_layer": dt2 / a.steps * 1e6 / a.layers,
                                             "tok_s": a.steps / dt2 * a.layers / LAYERS})
            except Exception as e:   # noqa: BLE001 -- the second leg must never cost the headline line
                oneshot_rec["status"] = f"unavailable: {type(e).__name__}: {e}"
                torch.cuda.synchronize()

    with emit_lock:
        if watchdog is not None:
            watchdog.cancel()
        emitted[0] = True
    if rank == 0:
        rec = build_rec()
        if world == 1 and not use_dist and not a.no_configs:
            rec["configs"] = other_configs(cfa, dev)
            rec["eager"] = eager_entries(dev)
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(S)
        print_rec(rec)
    if use_dist:
        # (after the line is out: a rank that died in an optional leg must not be able to hold the result back in here)
        bye = threading.Timer(20.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
