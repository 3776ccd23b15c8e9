"""Mint golden fixtures for tests/golden/ by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python oracle/gen_golden.py

What is imported from the reference (read-only, never copied):
  * tests/test_llama_tilelang.py : ``reference()``   -- the eager definition of the
    sglang variant (NEOX RoPE, residual add, [out,in] weights, eps argument).
  * chat/llama/model.py : ``RMSNorm``, ``precompute_freqs_cis``, ``apply_rotary_emb``,
    ``repeat_kv`` -- imported with stub modules standing in for the absent third-party
    packages it imports at module level (fairscale, flashinfer, clusterfusion); the stubs
    provide no arithmetic.  They pin the plain variant (GPT-J RoPE, eps 1e-6, [in,out]
    weights, caller-side cache handling of chat/llama/model.py:353-405).

Fixtures are DATA: the config, the seed, a checksum of the regenerated inputs, and the
fp16 outputs.  Inputs are re-drawn at test time by oracle.cf_oracle.make_inputs (weights
are 134 MB -- never committed).
"""
from __future__ import annotations

import importlib.util
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import cf_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _load_reference_eager():
    spec = importlib.util.spec_from_file_location(
        "ref_test_llama_tilelang", os.path.join(REF, "tests", "test_llama_tilelang.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.reference


def _load_reference_model():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    stub("fairscale")
    stub("fairscale.nn")
    stub("fairscale.nn.model_parallel")
    stub("fairscale.nn.model_parallel.initialize")
    stub("fairscale.nn.model_parallel.layers", ColumnParallelLinear=object,
         ParallelEmbedding=object, RowParallelLinear=object)
    stub("flashinfer")
    had_cf = sys.modules.get("clusterfusion")
    stub("clusterfusion", llama_decoder_layer=None)
    spec = importlib.util.spec_from_file_location(
        "ref_llama_model", os.path.join(REF, "chat", "llama", "model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if had_cf is not None:
        sys.modules["clusterfusion"] = had_cf
    else:
        del sys.modules["clusterfusion"]
    return mod


def _save(name, cfg, inp, outs):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(cfg)
    meta["input_sha256"] = O.input_checksum(inp)
    arrays = {k: v.detach().cpu().numpy() for k, v in outs.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **arrays)
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items()))


def gen_neox(reference):
    """sglang variant through the reference's own eager ``reference()``."""
    cases = [
        # (name, seed, S, distribution kwargs, eps)
        ("neox_s1_tl", 1, 1, dict(act_scale=1.0, kv_scale=1.0, angle_max=math.pi), 1e-5),
        ("neox_s37_tl", 2, 37, dict(act_scale=1.0, kv_scale=1.0, angle_max=math.pi), 1e-5),
        ("neox_s256_tl", 3, 256, dict(act_scale=1.0, kv_scale=1.0, angle_max=math.pi), 1e-5),
        ("neox_s128", 42, 128, dict(), 1e-6),
        ("neox_s1024", 42, 1024, dict(), 1e-6),
        ("neox_s4096", 42, 4096, dict(), 1e-6),
    ]
    for name, seed, S, dist, eps in cases:
        inp = O.make_inputs(seed, S, O.LLAMA2_7B, weight_layout="out_in", **dist)
        out, res, k, v = reference(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"],
                                   inp["k_cache"], inp["v_cache"], inp["rms_w"], eps,
                                   inp["cos"], inp["sin"])
        cfg = dict(variant="sglang", rope_style="neox", weight_layout="out_in", seed=seed,
                   seq_len=S, eps=eps, dist=dist, dims=[4096, 32, 32, 128],
                   source="reference tests/test_llama_tilelang.py:reference")
        _save(name, cfg, inp, dict(out=out, residual=res, k_new=k, v_new=v))


PAGED_CASES = [
    # (name, seed, page size, cached tokens per row): the reference's batched entry has no eager twin of its own
    # (kernel_batch_sglang.cuh:43-664; no test in the reference), so each row goes through the SAME ``reference()`` on the K/V
    # rows its page-table entries name -- what the kernel's per-sequence body computes (:118-122, :343-344)
    ("paged_p1_b6", 11, 1, [5, 333, 64, 1023, 2100, 1]),
    ("paged_p16_b3", 12, 16, [700, 17, 2049]),
    ("paged_p1_b20", 13, 1, [37 * i % 411 + 1 for i in range(20)]),
]


def gen_paged(reference):
    """The paged / batched sglang variant, row by row through the reference's own eager ``reference()``."""
    for name, seed, page_size, lens in PAGED_CASES:
        inp = O.make_paged_inputs(seed, page_size, lens)
        outs, ress, ks, vs = [], [], [], []
        for b, n_tok in enumerate(lens):
            ent = inp["kv_indices"][int(inp["kv_indptr"][b]):int(inp["kv_indptr"][b + 1])].long()
            if page_size == 1:
                slots = ent[:-1]
            else:
                t = torch.arange(n_tok)
                slots = ent[t // page_size] * page_size + (t % page_size)
            cs = inp["cos_sin"][n_tok]
            out, res, k, v = reference(inp["x"][b:b + 1], inp["residual"][b:b + 1], inp["weight_qkv"], inp["weight_o"],
                                       inp["k_cache"][slots], inp["v_cache"][slots], inp["rms_w"], 1e-6,
                                       cs[:64].contiguous(), cs[64:].contiguous())
            outs.append(out.view(1, -1))
            ress.append(res.view(1, -1))
            ks.append(k.reshape(1, -1))
            vs.append(v.reshape(1, -1))
        cfg = dict(variant="sglang paged batch", rope_style="neox", weight_layout="out_in", seed=seed, page_size=page_size,
                   lens=list(lens), eps=1e-6, dims=[4096, 32, 32, 128],
                   source="reference tests/test_llama_tilelang.py:reference, one call per row on the rows its page table names")
        _save(name, cfg, inp, dict(out=torch.cat(outs), residual=torch.cat(ress), k_new=torch.cat(ks), v_new=torch.cat(vs)))


def gen_gptj(model):
    """Plain variant: RMSNorm / RoPE / repeat_kv from the reference's chat/llama/model.py,
    composed exactly as its eager attention branch composes them (model.py:376-405), fed the
    fused-op call pattern of model.py:353-374 (un-normed x, cache[:start_pos], norm weight)."""
    D, H, hd = 4096, 32, 128
    for name, seed, S in [("gptj_s64", 7, 64), ("gptj_s1024", 42, 1024)]:
        inp = O.make_inputs(seed, S, O.LLAMA2_7B, weight_layout="in_out")
        start_pos = S
        freqs_cis = model.precompute_freqs_cis(hd, 2 * 4096)[start_pos:start_pos + 1]
        # what Attention.__init__ registers (model.py:276-282) and forward slices (:365-366)
        cos = torch.repeat_interleave(freqs_cis.real, 2, dim=-1).float()   # [1,128]
        sin = torch.repeat_interleave(freqs_cis.imag, 2, dim=-1).float()
        inp["cos"], inp["sin"] = cos.contiguous(), sin.contiguous()
        norm = model.RMSNorm(D, eps=1e-6)
        with torch.no_grad():
            norm.weight.copy_(inp["rms_w"].float())
            x = inp["x"].float().view(1, 1, D)
            xn = norm(x)
            w = inp["weight_qkv"].float()
            xq = (xn @ w[:D]).view(1, 1, H, hd)
            xk = (xn @ w[D:2 * D]).view(1, 1, H, hd)
            xv = (xn @ w[2 * D:]).view(1, 1, H, hd)
            xq, xk = model.apply_rotary_emb(xq, xk, freqs_cis=freqs_cis)
            keys = torch.cat([inp["k_cache"].float().view(1, S, H, hd), xk], 1)
            values = torch.cat([inp["v_cache"].float().view(1, S, H, hd), xv], 1)
            keys = model.repeat_kv(keys, 1).transpose(1, 2)
            values = model.repeat_kv(values, 1).transpose(1, 2)
            q = xq.transpose(1, 2)
            scores = torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(hd)
            scores = torch.softmax(scores.float(), dim=-1)
            o = torch.matmul(scores, values).transpose(1, 2).contiguous().view(1, D)
            out = o @ inp["weight_o"].float()
        cfg = dict(variant="plain", rope_style="gptj", weight_layout="in_out", seed=seed,
                   seq_len=S, eps=1e-6, dist={}, dims=[4096, 32, 32, 128], start_pos=start_pos,
                   source="reference chat/llama/model.py RMSNorm+apply_rotary_emb+eager attention")
        _save(name, cfg, inp, dict(out=out.half(), k_new=xk.half().view(1, H, hd),
                                   v_new=xv.half().view(1, H, hd), cos=cos, sin=sin))


def gen_gqa(model):
    """Grouped-query attention (BASELINE config 4: 32 q / 8 kv heads): the reference's fused kernels have no GQA path, its eager
    model does -- ``repeat_kv`` (chat/llama/model.py:166-175) says which kv head a q head reads.  RMSNorm / apply_rotary_emb /
    repeat_kv from model.py, composed as its eager attention branch composes them (model.py:376-405), [out,in] weights."""
    D, H, Hkv, hd = 4096, 32, 8, 128
    dims = O.LayerDims(D, H, Hkv, hd)
    for name, seed, S in [("gqa_gptj_s300", 21, 300), ("gqa_gptj_s2100", 22, 2100)]:
        inp = O.make_inputs(seed, S, dims, weight_layout="out_in")
        freqs_cis = model.precompute_freqs_cis(hd, 2 * 4096)[S:S + 1]
        cos = torch.repeat_interleave(freqs_cis.real, 2, dim=-1).float().contiguous()
        sin = torch.repeat_interleave(freqs_cis.imag, 2, dim=-1).float().contiguous()
        norm = model.RMSNorm(D, eps=1e-6)
        with torch.no_grad():
            norm.weight.copy_(inp["rms_w"].float())
            xn = norm(inp["x"].float().view(1, 1, D))
            w = inp["weight_qkv"].float()
            xq = (xn @ w[:H * hd].T).view(1, 1, H, hd)
            xk = (xn @ w[H * hd:(H + Hkv) * hd].T).view(1, 1, Hkv, hd)
            xv = (xn @ w[(H + Hkv) * hd:].T).view(1, 1, Hkv, hd)
            xq, xk = model.apply_rotary_emb(xq, xk, freqs_cis=freqs_cis)
            keys = torch.cat([inp["k_cache"].float().view(1, S, Hkv, hd), xk], 1)
            values = torch.cat([inp["v_cache"].float().view(1, S, Hkv, hd), xv], 1)
            keys = model.repeat_kv(keys, H // Hkv).transpose(1, 2)
            values = model.repeat_kv(values, H // Hkv).transpose(1, 2)
            q = xq.transpose(1, 2)
            scores = torch.softmax((torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(hd)).float(), dim=-1)
            o = torch.matmul(scores, values).transpose(1, 2).contiguous().view(1, H * hd)
            out = o @ inp["weight_o"].float().T
        cfg = dict(variant="plain, grouped-query", rope_style="gptj", weight_layout="out_in", seed=seed, seq_len=S, eps=1e-6,
                   dist={}, dims=[D, H, Hkv, hd], source="reference chat/llama/model.py RMSNorm + apply_rotary_emb + repeat_kv + eager attention")
        _save(name, cfg, inp, dict(out=out.half(), k_new=xk.half().view(1, Hkv, hd), v_new=xv.half().view(1, Hkv, hd), cos=cos, sin=sin))


GQA_PAGED_CASES = [("gqa_paged_p1_b3", 31, 1, [700, 17, 2100]), ("gqa_paged_p16_b2", 32, 16, [300, 4200])]


def gen_gqa_paged(model):
    """The grouped-query geometry (32 q / 8 kv heads) with SEVERAL sequences over a paged cache (configs 4 and f2 composed: the
    small-batch kernel of round 6).  Row by row through the reference's own model.py helpers, composed as in ``gen_gqa``: RMSNorm,
    apply_rotary_emb at the row's position (GPT-J pairs), repeat_kv, eager attention over the K/V rows the row's page-table entries
    name, then the O projection.  The fixture also keeps the rows' RoPE values (cos / sin of ``precompute_freqs_cis`` at each
    row's position): the test scatters them into the position-indexed tables the paged entry reads."""
    D, H, Hkv, hd = 4096, 32, 8, 128
    dims = O.LayerDims(D, H, Hkv, hd)
    table = model.precompute_freqs_cis(hd, 2 * 4096)
    for name, seed, page_size, lens in GQA_PAGED_CASES:
        inp = O.make_paged_inputs(seed, page_size, lens, dims)
        norm = model.RMSNorm(D, eps=1e-6)
        outs, ks, vs, coss, sins = [], [], [], [], []
        with torch.no_grad():
            norm.weight.copy_(inp["rms_w"].float())
            w = inp["weight_qkv"].float()
            for b, n_tok in enumerate(lens):
                ent = inp["kv_indices"][int(inp["kv_indptr"][b]):int(inp["kv_indptr"][b + 1])].long()
                if page_size == 1:
                    slots = ent[:-1]
                else:
                    t = torch.arange(n_tok)
                    slots = ent[t // page_size] * page_size + (t % page_size)
                freqs_cis = table[n_tok:n_tok + 1]
                xn = norm(inp["x"][b].float().view(1, 1, D))
                xq = (xn @ w[:H * hd].T).view(1, 1, H, hd)
                xk = (xn @ w[H * hd:(H + Hkv) * hd].T).view(1, 1, Hkv, hd)
                xv = (xn @ w[(H + Hkv) * hd:].T).view(1, 1, Hkv, hd)
                xq, xk = model.apply_rotary_emb(xq, xk, freqs_cis=freqs_cis)
                keys = torch.cat([inp["k_cache"][slots].float().view(1, n_tok, Hkv, hd), xk], 1)
                values = torch.cat([inp["v_cache"][slots].float().view(1, n_tok, Hkv, hd), xv], 1)
                keys = model.repeat_kv(keys, H // Hkv).transpose(1, 2)
                values = model.repeat_kv(values, H // Hkv).transpose(1, 2)
                q = xq.transpose(1, 2)
                scores = torch.softmax((torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(hd)).float(), dim=-1)
                o = torch.matmul(scores, values).transpose(1, 2).contiguous().view(1, H * hd)
                outs.append((o @ inp["weight_o"].float().T).half())
                ks.append(xk.half().view(1, Hkv * hd))
                vs.append(xv.half().view(1, Hkv * hd))
                coss.append(torch.repeat_interleave(freqs_cis.real, 2, dim=-1).float())
                sins.append(torch.repeat_interleave(freqs_cis.imag, 2, dim=-1).float())
        cfg = dict(variant="plain, grouped-query, paged batch", rope_style="gptj", weight_layout="out_in", seed=seed, page_size=page_size,
                   lens=list(lens), eps=1e-6, dims=[D, H, Hkv, hd],
                   source="reference chat/llama/model.py RMSNorm + apply_rotary_emb + repeat_kv + eager attention, one pass per row on the rows its page table names")
        _save(name, cfg, inp, dict(out=torch.cat(outs), k_new=torch.cat(ks), v_new=torch.cat(vs), cos=torch.cat(coss), sin=torch.cat(sins)))


def gen_helpers(model):
    """Tiny known-answer vectors for the RoPE / RMSNorm helpers themselves."""
    g = torch.Generator().manual_seed(123)
    t = torch.randn(1, 1, 4, 128, generator=g)
    fc = model.precompute_freqs_cis(128, 64)[17:18]
    rq, _ = model.apply_rotary_emb(t, t.clone(), freqs_cis=fc)
    x = torch.randn(3, 4096, generator=g)
    w = torch.randn(4096, generator=g)
    norm = model.RMSNorm(4096, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(w)
        y = norm(x)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "helpers.npz"),
                        rope_in=t.numpy(), rope_cos=fc.real.numpy(), rope_sin=fc.imag.numpy(),
                        rope_out=rq.numpy(), rms_in=x.numpy(), rms_w=w.numpy(), rms_out=y.numpy())
    print("wrote helpers")


if __name__ == "__main__":
    torch.set_num_threads(8)
    reference = _load_reference_eager()
    model = _load_reference_model()
    gen_helpers(model)
    gen_neox(reference)
    gen_paged(reference)
    gen_gptj(model)
    gen_gqa(model)
    gen_gqa_paged(model)
