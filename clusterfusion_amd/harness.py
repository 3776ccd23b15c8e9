"""Caller-side mirror of the reference's fused-attention call pattern.

Reproduces what /root/reference/chat/llama/model.py does around the op, so "drops into
chat/llama unchanged" can be exercised without fairscale / flashinfer / checkpoints:

  * weight prep  = Attention._build_cf_weights (model.py:292-328): weight_qkv =
    cat(wq.T, wk.T, wv.T) [12288,4096], weight_o = wo.T, fp16;
  * rotary tables = model.py:276-282: precompute_freqs_cis(head_dim, 2*max_seq_len) with
    real/imag repeat_interleave(2) -> [L,128] fp32 (GPT-J pairs);
  * decode step  = Attention.forward fused branch (model.py:353-374): pass the un-normed x,
    cache[:start_pos] views, attention_norm.weight, cos/sin[start_pos]; write the returned k/v
    at start_pos; the residual add is the caller's (model.py:492).
"""
from __future__ import annotations

import torch

from .ops import llama_decoder_layer


def precompute_rotary(head_dim: int, end: int, theta: float = 10000.0, device="cpu"):
    """cos/sin tables [end, head_dim] fp32, pair-duplicated (model.py:96-121 + :276-282)."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(end).float(), freqs)
    cos = torch.repeat_interleave(torch.cos(ang), 2, dim=-1)
    sin = torch.repeat_interleave(torch.sin(ang), 2, dim=-1)
    return cos.to(device), sin.to(device)


class FusedAttentionBlock:
    """Llama-2-7B attention block of ONE layer, decode only, driven through the drop-in op."""

    def __init__(self, wq, wk, wv, wo, attention_norm_weight, max_seq_len: int, op=llama_decoder_layer):
        dim = wq.shape[0]
        dev = wq.device
        w = torch.empty(3 * dim, dim, device=dev, dtype=torch.float16)
        w[:dim] = wq.t()
        w[dim:2 * dim] = wk.t()
        w[2 * dim:] = wv.t()
        self.weight_qkv = w.contiguous()
        self.weight_o = wo.t().contiguous().to(torch.float16)
        self.norm_weight = attention_norm_weight.to(torch.float16).contiguous()
        self.n_heads, self.head_dim, self.dim = 32, dim // 32, dim
        self.cache_k = torch.zeros(1, max_seq_len, self.n_heads, self.head_dim, device=dev, dtype=torch.float16)
        self.cache_v = torch.zeros_like(self.cache_k)
        self.rotary_cos, self.rotary_sin = precompute_rotary(self.head_dim, max_seq_len * 2, device=dev)
        self.op = op

    def forward(self, x: torch.Tensor, start_pos: int) -> torch.Tensor:
        """x [1,1,dim] un-normed; returns x + attention(x) like clusterfusion_forward (model.py:488-492)."""
        bsz, seqlen, _ = x.shape
        kv_k = self.cache_k[:bsz, :start_pos].view(-1, self.n_heads * self.head_dim)
        kv_v = self.cache_v[:bsz, :start_pos].view(-1, self.n_heads * self.head_dim)
        out, xk, xv = self.op(x, self.weight_qkv, self.weight_o, kv_k, kv_v, self.norm_weight,
                              self.rotary_cos[start_pos:start_pos + seqlen],
                              self.rotary_sin[start_pos:start_pos + seqlen])
        self.cache_k[:bsz, start_pos:start_pos + seqlen] = xk
        self.cache_v[:bsz, start_pos:start_pos + seqlen] = xv
        return x + out.view(bsz, seqlen, self.dim)
