"""CPU, world_size 2, gloo: the N>1 path of the decode op = shard by head + local op + ONE
all-reduce(sum) of the fp16 O-projection partial (clusterfusion_amd/tp.py).  The local op is
injected: here the oracle stands in for the HIP kernel (which needs a GPU), so what is under test is
the host logic that bench.py --gpus N and a TP caller run: shard_layer_weights / shard_kv_cache /
decoder_layer_tp and the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_op(x, residual, wqkv, wo, kc, vc, rms_w, eps, cos, sin, *, n_q_heads, n_kv_heads, head_dim,
                     weight_layout="out_in", rope_style="neox"):
    dims = O.LayerDims(rms_w.numel(), n_q_heads, n_kv_heads, head_dim)
    return O.decoder_layer(x, residual, wqkv, wo, kc, vc, rms_w, eps, cos, sin, dims=dims,
                           weight_layout=weight_layout, rope_style=rope_style)


def _worker(rank, world, port, layout, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clusterfusion_amd.tp import ShardSpec, decoder_layer_tp, shard_kv_cache, shard_layer_weights
        torch.set_num_threads(2)
        dims = O.LayerDims(1024, 8, 8, 128)
        inp = O.make_inputs(17, 45, dims, weight_layout=layout)
        style = "neox" if layout == "out_in" else "gptj"
        if style == "gptj":
            inp["cos"] = inp["cos"].repeat_interleave(2)
            inp["sin"] = inp["sin"].repeat_interleave(2)
        spec = ShardSpec(1024, 8, 8, 128, rank, world)
        w, wo = shard_layer_weights(inp["weight_qkv"], inp["weight_o"], spec, layout)
        kc, vc = shard_kv_cache(inp["k_cache"], spec), shard_kv_cache(inp["v_cache"], spec)
        out, res, k, v = decoder_layer_tp(_oracle_local_op, spec, None, inp["x"], inp["residual"], w, wo, kc, vc,
                                          inp["rms_w"], 1e-6, inp["cos"], inp["sin"],
                                          weight_layout=layout, rope_style=style)
        full = O.decoder_layer(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"], inp["k_cache"],
                               inp["v_cache"], inp["rms_w"], 1e-6, inp["cos"], inp["sin"], dims=dims,
                               weight_layout=layout, rope_style=style)
        err = (out.float() - full[0].float()).abs().max().item()
        # k_new stays rank-local: this rank's heads of the full k
        hk = spec.local_kv_heads
        kf = full[2][:, rank * hk:(rank + 1) * hk]        # BLAS blocking differs with the shard shape: <= 1 ulp
        k_ok = (k.float() - kf.float()).abs().max().item() <= 2.0 ** -10 * max(1.0, kf.float().abs().max().item())
        # every rank holds the same reduced output
        gathered = [torch.empty_like(out) for _ in range(world)]
        dist.all_gather(gathered, out)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        q.put((rank, err, k_ok, same, torch.equal(res, full[1])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["out_in", "in_out"])
def test_tp2_gloo_allreduce_matches_unsharded(layout):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, k_ok, same, res_ok in res:
        assert err <= 2e-3, (rank, err)          # two fp16-rounded partials summed in fp16
        assert k_ok and same and res_ok


def test_pluggable_reducer_replaces_the_collective():
    """decoder_layer_tp(reducer=...) hands the partial to the caller's reducer instead of dist.all_reduce (world 1 here: no
    process group needed); clusterfusion_amd.tp.OneShotReducer is such a reducer on the GPU."""
    from clusterfusion_amd.tp import ShardSpec, decoder_layer_tp
    dims = O.LayerDims(1024, 8, 8, 128)
    inp = O.make_inputs(3, 9, dims)
    spec = ShardSpec(1024, 8, 8, 128, 0, 1)
    seen = []

    def reducer(out):
        seen.append(out.clone())
        return out * 2

    out, res, k, v = decoder_layer_tp(_oracle_local_op, spec, None, inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"],
                                      inp["k_cache"], inp["v_cache"], inp["rms_w"], 1e-6, inp["cos"], inp["sin"], reducer=reducer)
    assert len(seen) == 1 and torch.equal(out, seen[0] * 2)


def test_shard_spec_validation():
    from clusterfusion_amd.tp import ShardSpec
    with pytest.raises(ValueError):
        ShardSpec(4096, 32, 32, 128, 0, 5)
    with pytest.raises(ValueError):
        ShardSpec(4096, 32, 32, 128, 0, 16)       # 2 local heads: not a 512-wide strip
    s = ShardSpec(4096, 32, 8, 128, 1, 2)
    assert s.local_q_heads == 16 and s.local_kv_heads == 4


def test_bench_self_spawns_its_ranks_dry_launch():
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE in the environment) must launch its own ranks.  --dry-launch
    runs that launch path without a GPU: two processes through torch.distributed.run on 127.0.0.1, a gloo rendezvous, one
    all-reduce, ONE JSON line from rank 0, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec == {"dry_launch": True, "n_gpus": 2, "gpus_arg": 2, "all_reduce_sum": 3.0, "expected": 3.0}


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in the build container: `bench.py --gpus 2` (a real run) must say so and exit non-zero, not hang or fall back."""
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=root)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
