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

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HIDDEN, HEADS, HEAD_DIM, LAYERS = 4096, 32, 128, 32
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md); ~6290 measured copy


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
    ap.add_argument("--kv-splits", type=int, default=0)
    ap.add_argument("--path", default="auto", choices=["auto", "pipeline", "fused"])
    ap.add_argument("--debug-flags", type=int, default=0, help="experiment bits for the fused kernel (cf_debug_set_flags)")
    return ap.parse_args()


def build_layers(cfa, dev, world, rank, n_layers, S, page_size, seed=42):
    """Per-layer synthetic state on the device + one PreparedLayer per layer."""
    g = torch.Generator(device=dev).manual_seed(seed + rank)
    hq = HEADS // world
    qd = hq * HEAD_DIM

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
    layers = []
    x, res = x0, res0
    for li in range(n_layers):
        w_qkv = rn(3 * qd, HIDDEN)
        w_o = rn(HIDDEN, qd)
        rms_w = rn(HIDDEN)
        kc = rn(pool_pages * page_size, qd)
        vc = rn(pool_pages * page_size, qd)
        perm = torch.randperm(pool_pages, generator=g, device=dev)[:n_pages_need].to(torch.int32).contiguous()
        out = torch.empty(1, HIDDEN, dtype=torch.float16, device=dev)
        res_out = torch.empty(1, HIDDEN, dtype=torch.float16, device=dev)
        p = cfa.prepare_decoder_layer(
            x, res, w_qkv, w_o, kc, vc, rms_w, 1e-6, cos_sin, cos_sin.view(-1)[HEAD_DIM // 2:],
            n_q_heads=hq, n_kv_heads=hq, kv_indptr=indptr, kv_indices=perm, kv_seq_lens=seq_lens,
            page_size=page_size, max_seq_len=S, positions=positions, rope_row_stride=HEAD_DIM,
            out=out, residual_out=res_out, write_kv_to_cache=True, want_kv=False)
        layers.append(p)
        # chain like a model: next layer's input = this layer's output, residual = updated residual
        # (the FFN half between them is outside the fused op: chat/llama/model.py:519)
        x, res = out, res_out
    return layers


def cpu_baseline(S, budget_s=20.0):
    """The oracle (oracle/cf_oracle.py, a PyTorch-CPU port of the reference's eager layer,
    tests/test_llama_tilelang.py:18-49) timed on this box's host cores for ONE layer of the same
    workload, weights pre-converted to fp32 once.  Reported baseline, not a target."""
    from oracle import cf_oracle as O
    inp = O.make_inputs(42, S, O.LLAMA2_7B)
    f = {k: (v.float() if v.dtype == torch.float16 else v) for k, v in inp.items()}
    args = (f["x"], f["residual"], f["weight_qkv"], f["weight_o"], f["k_cache"], f["v_cache"], f["rms_w"],
            1e-6, f["cos"], f["sin"])
    O.decoder_layer(*args)
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 200:
        t0 = time.perf_counter()
        O.decoder_layer(*args)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / (LAYERS * med), "unit": "tok/s", "cores": torch.get_num_threads(), "kind": "port",
            "us_per_layer": med * 1e6,
            "sample": f"{len(times)} calls of one Llama-2-7B layer bs=1 seq={S} (oracle, fp32 weights pre-converted, "
                      f"contiguous KV), median; host cpu_count={os.cpu_count()}"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CF_BENCH_TP=N (debug): run the per-rank workload of an N-way head-parallel shard on however many
    # ranks were launched (lets a 1-GPU box exercise the TP code path incl. the RCCL all-reduce)
    tp = int(os.environ.get("CF_BENCH_TP", str(world)))
    force_dist = os.environ.get("CF_BENCH_FORCE_DIST", "0") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    import clusterfusion_amd as cfa
    if a.kv_splits:
        cfa.set_tuning(a.kv_splits)
    if a.debug_flags:
        from clusterfusion_amd import _lib
        _lib.load().cf_debug_set_flags(a.debug_flags)
    cfa.set_path(a.path)
    S = a.seq
    layers = build_layers(cfa, dev, tp, rank, a.layers, S, a.page_size)
    outs = [p.outputs[0] for p in layers]

    def step():
        for p, o in zip(layers, outs):
            p.run()
            if use_dist:
                dist.all_reduce(o)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()        # synthetic state was drawn on the default stream
    stream = torch.cuda.Stream(dev)
    graph = None
    with torch.cuda.stream(stream):
        step()                      # first call: lazy init (workspace, RCCL channels)
        torch.cuda.synchronize()
        if not a.no_graph:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    step()
                graph = g
            except Exception as e:   # noqa: BLE001 -- capture is an optimisation of the host side only
                print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        run = graph.replay if graph is not None else step
        for _ in range(a.warmup):
            run()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)          # HIP events on the stream the kernels are launched on
        for _ in range(a.steps):
            run()
        ev1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)

        # per-kernel durations: HIP events recorded by the library on ITS launch stream around each
        # kernel, eager launches, same workload (a separate pass so the events do not sit in `dt`)
        stage_ms, ncalls = [0.0] * 4, 0
        if rank == 0:
            cfa.profile_enable(True)
            for _ in range(max(2, min(a.steps, 20))):
                for p in layers:
                    p.run()
            torch.cuda.synchronize()
            stage_ms, ncalls = cfa.profile_read(reset=True)
            cfa.profile_enable(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        us_layer = ms_per_step * 1e3 / a.layers
        hq = HEADS // tp
        bytes_layer = cfa.algorithmic_bytes(S, HIDDEN, hq, hq, HEAD_DIM, 1, True)
        stage_us = [m * 1e3 / max(ncalls, 1) for m in stage_ms]
        path = cfa.last_path()
        if path == "fused":
            # ONE persistent kernel per layer: its algorithmic bytes are the layer's
            # duration: HIP-event pair around the timed region on the launch stream / number of launches
            # (the launches are back to back, so this includes the ~0.3 us inter-launch gap; events
            # recorded BETWEEN launches would add their own ~4 us of command-processor gap each)
            kern = "k_fused_decode_mha" if tp == 1 else f"k_fused_decode_g<{hq},1>"   # head-parallel shard: hq local heads
            kern_name, kern_bytes = kern + " (whole layer, one persistent launch)", bytes_layer
            kern_us = ev_ms * 1e3 / (a.steps * a.layers)
            if use_dist:   # the timed region also holds the all-reduce: take the kernel alone (library events)
                kern_us = stage_us[0]
        else:
            # dominant kernel = stage 0 (RMSNorm + QKV projection): Wqkv shard + x, residual, rms_w, raw q|k|v out
            kern_name = "k_qkv_rows (RMSNorm + QKV GEMV)"
            kern_bytes = 2 * HIDDEN * 3 * hq * HEAD_DIM + 3 * 2 * HIDDEN + 4 * 3 * hq * HEAD_DIM
            kern_us = stage_us[0]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kern_name.split(" ")[0] + "_bytes_per_launch")
            except Exception:   # noqa: BLE001
                traffic = None
        roof = {"bound": "hbm", "kernel": kern_name,
                "achieved": kern_bytes / (kern_us * 1e-6) / 1e9 if kern_us > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (kern_bytes / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if kern_us > 0 else None,
                "traffic": traffic, "bytes_per_launch": kern_bytes, "us_per_launch": kern_us,
                "timing": ("hipEvents recorded by the library around the kernel, eager launches (the timed region also holds the all-reduce)"
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
        rec = {
            "metric": "decode tok/s through the fused attention-block op of 32 layers (us/decoder-layer alongside), "
                      "Llama-2-7B bs=1 seq=4096",
            "value": 1e3 / (ms_per_step * LAYERS / a.layers),
            "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "us_per_layer": us_layer, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f16 storage, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"Llama-2-7B fused attention-block decode, bs=1 seq={S}, paged KV page_size="
                                   f"{a.page_size}, {a.layers} distinct layers per step (BASELINE configs[2]"
                                   + (f" sharded head-parallel TP={world}, RCCL all-reduce per layer = configs[4])"
                                      if world > 1 else ")"),
                       "parallelism": f"tp{tp}", "launch": "hipGraph replay" if graph is not None else "eager",
                       "kv_splits": a.kv_splits or "auto", "path": path},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(S)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the LAST line
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
