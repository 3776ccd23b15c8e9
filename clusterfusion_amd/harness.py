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


class DecodeModel:
    """A whole Llama-style decoder for single-sequence greedy decode: every layer's attention
    block is ONE call of the fused op (paged entry, the op writes the new K/V into the cache itself), the two
    RMSNorms between blocks are ``clusterfusion.rmsnorm`` in its fused-add form, and the SwiGLU FFN / LM head are
    plain ``torch`` matmuls (rocBLAS / hipBLASLt) -- that half of a layer is outside the reference's fused op too
    (chat/llama/model.py:451-520).  Random weights of the right shapes; the point is a whole-model tokens/s with the
    fused op in its real place, replayed from one HIP graph per token.

    State lives on the device so a step can be graph-captured: ``pos`` (int64[1]), ``indptr`` (int32[2]), the
    current token id.  ``step()`` = embed -> n_layers x (attention block, add+norm, FFN) -> add+norm -> LM head ->
    argmax -> advance."""

    def __init__(self, n_layers=32, hidden=4096, n_heads=32, n_kv_heads=None, ffn=11008, vocab=32000, max_seq=8192,
                 start_pos=4096, device="cuda:0", seed=0, w_scale=0.02):
        from . import ops
        self.ops = ops
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        n_kv = n_kv_heads or n_heads
        hd = 128
        self.hidden, self.n_layers, self.n_heads, self.n_kv, self.max_seq = hidden, n_layers, n_heads, n_kv, max_seq

        def rn(*shape, scale=w_scale):
            return (torch.randn(*shape, generator=g, device=dev) * scale).half()

        self.embed = rn(vocab, hidden, scale=1.0)
        self.lm_head = rn(vocab, hidden)
        self.final_norm = (1 + rn(hidden, scale=0.1))
        self.layers = []
        for _ in range(n_layers):
            self.layers.append(dict(
                wqkv=rn((n_heads + 2 * n_kv) * hd, hidden), wo=rn(hidden, n_heads * hd),
                attn_norm=1 + rn(hidden, scale=0.1), ffn_norm=1 + rn(hidden, scale=0.1),
                w_gate_up=rn(2 * ffn, hidden), w_down=rn(hidden, ffn),
                kc=rn(max_seq, n_kv * hd, scale=0.3), vc=rn(max_seq, n_kv * hd, scale=0.3)))
        self.kptrs = torch.tensor([l["kc"].data_ptr() for l in self.layers], dtype=torch.uint64, device=dev)
        self.vptrs = torch.tensor([l["vc"].data_ptr() for l in self.layers], dtype=torch.uint64, device=dev)
        cos, sin = precompute_rotary(hd, max_seq, device=dev)                 # [L,128] pair-duplicated
        self.cos_sin = torch.cat([cos[:, ::2], sin[:, ::2]], dim=1).contiguous().float()   # NEOX row: cos[64] | sin[64]
        # device-side decode state: token slots are the positions themselves (identity page table)
        self.indices = torch.arange(max_seq, dtype=torch.int32, device=dev)
        self.pos = torch.tensor([start_pos], dtype=torch.int64, device=dev)
        self.indptr = torch.tensor([0, start_pos + 1], dtype=torch.int32, device=dev)
        self.token = torch.zeros(1, dtype=torch.int64, device=dev)
        self.x = torch.empty(1, hidden, dtype=torch.float16, device=dev)
        self.res = torch.zeros(1, hidden, dtype=torch.float16, device=dev)
        self.attn_out = torch.empty(1, hidden, dtype=torch.float16, device=dev)
        self.ffn = ffn
        self.dev = dev

    def step(self):
        ops = self.ops
        torch.index_select(self.embed, 0, self.token, out=self.x)
        self.res.zero_()
        for li, L in enumerate(self.layers):
            # attention block: res := x + res (in place), attn_out := Attn(RMSNorm(res)); new K/V -> cache slot pos
            if self.n_kv == self.n_heads and self.hidden == 4096 and self.n_heads == 32:
                ops.llama_decoder_layer_batch_decode_sglang(
                    self.attn_out, self.res, self.x, self.res, L["wqkv"], L["wo"], self.indptr, self.indices,
                    self.kptrs, self.vptrs, li, L["attn_norm"], 1e-5, self.pos, self.cos_sin)
            else:
                ops.decoder_layer(self.x, self.res, L["wqkv"], L["wo"], None, None, L["attn_norm"], 1e-5,
                                  self.cos_sin, self.cos_sin.view(-1)[64:], n_q_heads=self.n_heads, n_kv_heads=self.n_kv,
                                  kv_indptr=self.indptr, kv_indices=self.indices, kv_cache_ptrs=(self.kptrs, self.vptrs),
                                  layer_id=li, positions=self.pos, rope_row_stride=128, out=self.attn_out,
                                  residual_out=self.res, write_kv_to_cache=True, max_seq_len=self.max_seq - 1,
                                  want_kv=False)
            # FFN block: res := attn_out + res; h = RMSNorm(res); x := W_down(silu(gate) * up)
            h = ops.rmsnorm(self.attn_out, L["ffn_norm"], 1e-5, residual=self.res, residual_out=self.res)
            # (torch.mv: rocBLAS' GEMV streams these weights at 4.6 / 3.0 TB/s where F.linear's GEMM path reaches 3.7 / 2.9 -- the FFN
            #  is outside the reference's fused op and outside this library; only its share of the tok/s figure is at stake)
            gu = torch.mv(L["w_gate_up"], h.view(-1))
            act = torch.nn.functional.silu(gu[: self.ffn]) * gu[self.ffn:]
            torch.mv(L["w_down"], act, out=self.x.view(-1))
        h = ops.rmsnorm(self.x, self.final_norm, 1e-5, residual=self.res, residual_out=self.res)
        logits = torch.mv(self.lm_head, h.view(-1)).view(1, -1)
        self.token.copy_(torch.argmax(logits, dim=-1))
        self.pos.add_(1)
        self.indptr[1:].add_(1)
        return logits
