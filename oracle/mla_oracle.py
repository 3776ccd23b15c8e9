"""CPU oracle for the DeepSeek-V2-Lite MLA decoder-layer (attention block) decode op.

TEST INFRASTRUCTURE ONLY (same rule as oracle/cf_oracle.py): nothing under ``clusterfusion_amd/``
may import this module; only ``tests/``, ``tools/`` benches acting as checkers and
``__graft_entry__`` do.

**Parity unpinned.**  The reference ships this op (pybind ``deepseek_decoder_layer``,
include/pybind.cpp:45-59,113) with NO test, golden vector or eager twin anywhere in the repo, and the
kernel itself (TMA + thread-block clusters, sm_90a) cannot be built or run in this image.  This file
is therefore a restatement of the CUDA kernel's arithmetic read off the source
(include/H100/deepseek/kernel.cuh, config.h, deepseek_kernel_dispatch.cu); it is cross-checked only
against an independently written "un-absorbed" MLA formula (tests/test_mla_oracle.py), which pins the
algebra but not the reference's outputs.

What the kernel computes (kernel.cuh line numbers), shapes from deepseek_kernel_dispatch.cu:55-206:

    x[2048] fp16, rms_input_weight[2048], eps = 1e-6 (:47)
    xn      = fp16(x * rsqrt(mean(x^2) + eps) * w)                                   :80-125
    q_nope  = xn @ weight_q_nope [2048, 16*128]     (head h = columns 128h..)        :127-161
    q_pe    = xn @ weight_q_pe   [2048, 16*64]                                       :163-206
    ckv     = xn @ weight_kv_nope[2048, 512]                                         :208-243
    k_pe    = xn @ weight_k_pe   [2048, 64]                                          :245-287
      (each is the fp16 sum of four K-quarter partials, one per CTA of the cluster   :289-296)
    q_pe, k_pe <- RoPE (rotate-half over 64 dims, fp32 cos/sin[64])                  :298-315
      -- computed but NEVER USED by the rest of the kernel: the attention below only touches the
         first 512 columns of the cache (TMA boxes at column 0 and 256, :407-408) and the 512-wide
         absorbed query.  ``rope_scores=False`` (default) reproduces that; ``rope_scores=True`` is
         the complete MLA score  q_abs . ckv + q_pe_rot . k_pe  (an extension of this repo).
    ckv_n   = fp16(ckv * rsqrt(mean(ckv^2) + eps) * rms_ckv_weight)                  :317-347
    q_abs[h]= q_nope[h] @ weight_uk[128, 16*512][:, 512h:512h+512]                   :349-394
    scores  : t < S-1: q_abs[h] . ckv_cache[t, :512];  the LAST cache row (t = S-1) is not read:
              the new token's ckv_n takes its place                                   :470-473
              scale = rsqrt(192) (:47), softmax over the S entries, running max starts at 0.0 (:47)
    a[h]    = sum_t p_t * ckv_cache[t, :512]  (+ p_new * ckv_n)   -> fp16             :459-581
    o_h[h]  = a[h] @ weight_uv[512, 16*128][:, 128h:128h+128]     -> fp16             :590-631
    out     = concat_h(o_h) @ weight_o[2048, 2048]  (fp16 atomics over heads)        :640-696

The oracle computes in float64 and rounds only where ``emulate_kernel_rounding`` asks for it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

__all__ = ["MlaDims", "DSV2_LITE", "mla_decoder_layer", "mla_decoder_layer_unabsorbed", "make_mla_inputs",
           "mla_algorithmic_bytes", "rope_half"]


@dataclass(frozen=True)
class MlaDims:
    """config.h:2-9"""
    hidden: int = 2048
    n_heads: int = 16
    nope: int = 128
    rope: int = 64
    kv_lora: int = 512

    @property
    def latent(self) -> int:          # MLA_HEAD_DIM
        return self.kv_lora + self.rope

    @property
    def qk_head_dim(self) -> int:     # HEAD_DIM, the softmax scale's dimension
        return self.nope + self.rope


DSV2_LITE = MlaDims()


def _r16(t: torch.Tensor) -> torch.Tensor:
    return t.half().to(t.dtype)


def rope_half(v: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """kernel.cuh:298-315 (q side): out[i] = v[i] cos[i] - v[i+32] sin[i+32]  (i < 32),
    out[i] = v[i] cos[i] + v[i-32] sin[i-32]  (i >= 32); v = [..., 64]."""
    half = v.shape[-1] // 2
    lo, hi = v[..., :half], v[..., half:]
    return torch.cat([lo * cos[:half] - hi * sin[half:], hi * cos[half:] + lo * sin[:half]], dim=-1)


def _split_k_fp16(x: torch.Tensor, w: torch.Tensor, parts: int) -> torch.Tensor:
    """x @ w as `parts` K-slices, each rounded to fp16 and added in fp16 (cluster_reduce, :289-296)."""
    k = x.shape[-1] // parts
    acc = None
    for p in range(parts):
        part = _r16(x[..., p * k:(p + 1) * k] @ w[p * k:(p + 1) * k])
        acc = part if acc is None else _r16(acc + part)
    return acc


def mla_decoder_layer(inp: Dict[str, torch.Tensor], dims: MlaDims = DSV2_LITE, *, eps: float = 1e-6,
                      rope_scores: bool = False, emulate_kernel_rounding: bool = False) -> Dict[str, torch.Tensor]:
    """Returns {"o": [1, hidden], "latent": [latent] = cat(ckv_n, k_pe_rot)} in float64.

    inp: the 13 tensors of the reference entry (deepseek_kernel_dispatch.cu:4-18); ckv_cache is [S, 576] whose
    last row is the slot of the new token (never read)."""
    f = lambda t: t.detach().double().cpu()
    rd = _r16 if emulate_kernel_rounding else (lambda t: t)
    H, N, R, L = dims.n_heads, dims.nope, dims.rope, dims.kv_lora
    x = f(inp["input"]).reshape(-1)
    assert x.numel() == dims.hidden
    xn = rd(x * torch.rsqrt((x * x).mean() + eps) * f(inp["rms_input_weight"]))
    proj = (lambda v, w: _split_k_fp16(v, w, 4)) if emulate_kernel_rounding else (lambda v, w: v @ w)
    q_nope = proj(xn, f(inp["weight_q_nope"])).reshape(H, N)
    q_pe = proj(xn, f(inp["weight_q_pe"])).reshape(H, R)
    ckv = proj(xn, f(inp["weight_kv_nope"]))
    k_pe = proj(xn, f(inp["weight_k_pe"]))
    cos, sin = f(inp["cos"]).reshape(-1)[:R], f(inp["sin"]).reshape(-1)[:R]
    q_pe_rot, k_pe_rot = rd(rope_half(q_pe, cos, sin)), rd(rope_half(k_pe, cos, sin))
    ckv_n = rd(ckv * torch.rsqrt((ckv * ckv).mean() + eps) * f(inp["rms_ckv_weight"]))
    w_uk = f(inp["weight_uk"]).reshape(N, H, L)
    q_abs = rd(torch.einsum("hn,nhl->hl", q_nope, w_uk))            # [H, 512]
    cache = f(inp["ckv_cache"])
    S = cache.shape[0]
    assert S >= 1 and cache.shape[1] == dims.latent
    keys = torch.cat([cache[:S - 1, :L], ckv_n[None]], dim=0)       # [S, 512]: new token replaces the last row
    scores = q_abs @ keys.T                                          # [H, S]
    if rope_scores:
        kpe = torch.cat([cache[:S - 1, L:], k_pe_rot[None]], dim=0)
        scores = scores + q_pe_rot @ kpe.T
    scores = scores / math.sqrt(dims.qk_head_dim)
    m = torch.clamp(scores.max(dim=-1, keepdim=True).values, min=0.0)   # running max starts at 0.0 (:47)
    p = torch.exp(scores - m)
    a = rd((p @ keys) / p.sum(dim=-1, keepdim=True))                # [H, 512]
    w_uv = f(inp["weight_uv"]).reshape(L, H, N)
    o_h = torch.einsum("hl,lhn->hn", a, w_uv)
    o_h = rd(o_h)
    w_o = f(inp["weight_o"])
    if emulate_kernel_rounding:                                     # fp16 atomics over heads (:680,695), fixed order here
        out = torch.zeros(dims.hidden, dtype=torch.float64)
        for h in range(H):
            out = _r16(out + _r16(o_h[h] @ w_o[h * N:(h + 1) * N]))
    else:
        out = o_h.reshape(-1) @ w_o
    return {"o": out.reshape(1, dims.hidden), "latent": torch.cat([ckv_n, k_pe_rot])}


def mla_decoder_layer_unabsorbed(inp: Dict[str, torch.Tensor], dims: MlaDims = DSV2_LITE, *, eps: float = 1e-6,
                                 rope_scores: bool = False) -> torch.Tensor:
    """Independent formula (the textbook MLA decode): expand every cached latent to per-head keys and values
    (k_h = ckv W_uk_h^T, v_h = ckv W_uv_h) and run ordinary attention.  Pins the absorbed algebra above."""
    f = lambda t: t.detach().double().cpu()
    H, N, R, L = dims.n_heads, dims.nope, dims.rope, dims.kv_lora
    x = f(inp["input"]).reshape(-1)
    xn = x / torch.sqrt((x * x).mean() + eps) * f(inp["rms_input_weight"])
    q_nope = (xn @ f(inp["weight_q_nope"])).reshape(H, N)
    q_pe = (xn @ f(inp["weight_q_pe"])).reshape(H, R)
    ckv = xn @ f(inp["weight_kv_nope"])
    ckv_n = ckv / torch.sqrt((ckv * ckv).mean() + eps) * f(inp["rms_ckv_weight"])
    cache = f(inp["ckv_cache"])
    S = cache.shape[0]
    lat = torch.cat([cache[:S - 1, :L], ckv_n[None]], dim=0)                       # [S, L]
    w_uk = f(inp["weight_uk"]).reshape(N, H, L)
    w_uv = f(inp["weight_uv"]).reshape(L, H, N)
    out_heads = []
    cos, sin = f(inp["cos"]).reshape(-1)[:R], f(inp["sin"]).reshape(-1)[:R]
    for h in range(H):
        k_h = lat @ w_uk[:, h, :].T                                                # [S, N]
        v_h = lat @ w_uv[:, h, :]                                                  # [S, N]
        s = k_h @ q_nope[h]
        if rope_scores:
            k_pe_new = rope_half(xn @ f(inp["weight_k_pe"]), cos, sin)
            kpe = torch.cat([cache[:S - 1, L:], k_pe_new[None]], dim=0)
            s = s + kpe @ rope_half(q_pe[h], cos, sin)
        pr = torch.softmax(s / math.sqrt(dims.qk_head_dim), dim=0)
        out_heads.append(pr @ v_h)
    return (torch.cat(out_heads) @ f(inp["weight_o"])).reshape(1, dims.hidden)


def make_mla_inputs(seed: int, seq_len: int, dims: MlaDims = DSV2_LITE, *, device: str = "cpu",
                    score_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded inputs with the reference entry's shapes and dtypes; magnitudes chosen so that every intermediate
    is O(1) (scores have a standard deviation of about `score_gain`)."""
    g = torch.Generator().manual_seed(seed)
    H, N, R, L, D = dims.n_heads, dims.nope, dims.rope, dims.kv_lora, dims.hidden
    rn = lambda *s: torch.randn(*s, generator=g)
    inp = {
        "input": rn(1, D),
        "weight_q_nope": rn(D, H * N) / math.sqrt(D),
        "weight_q_pe": rn(D, H * R) / math.sqrt(D),
        "weight_uk": rn(N, H * L) / math.sqrt(N) * score_gain * math.sqrt(dims.qk_head_dim / L),
        "weight_kv_nope": rn(D, L) / math.sqrt(D),
        "weight_k_pe": rn(D, R) / math.sqrt(D),
        "weight_uv": rn(L, H * N) / math.sqrt(L),
        "weight_o": rn(H * N, D) / math.sqrt(H * N),
        "ckv_cache": rn(seq_len, dims.latent),
        "rms_input_weight": 1.0 + 0.1 * rn(D),
        "rms_ckv_weight": 1.0 + 0.1 * rn(L),
    }
    inp = {k: v.half() for k, v in inp.items()}
    pos = float(seq_len - 1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, R, 2, dtype=torch.float64) / R))
    ang = torch.cat([pos * inv, pos * inv])
    inp["cos"], inp["sin"] = torch.cos(ang).float(), torch.sin(ang).float()
    return {k: v.to(device) for k, v in inp.items()}


def mla_algorithmic_bytes(dims: MlaDims, seq_len: int, rope_scores: bool = False) -> int:
    """Every weight byte once + the cached latents that are read + vectors."""
    H, N, R, L, D = dims.n_heads, dims.nope, dims.rope, dims.kv_lora, dims.hidden
    w = D * H * N + N * H * L + D * L + L * H * N + H * N * D
    if rope_scores:                                        # q_pe / k_pe only matter when they enter the scores
        w += D * H * R + D * R
    cache = max(seq_len - 1, 0) * (dims.latent if rope_scores else L)
    vec = D + D + L + D                                    # x, rms weights, out
    return 2 * (w + cache + vec) + 4 * 2 * R
