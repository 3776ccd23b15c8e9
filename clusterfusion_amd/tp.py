"""Head-parallel tensor parallelism for the fused decode op (BASELINE config 5).

One process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm).  Rank r
owns q heads [r*Hq/w, (r+1)*Hq/w) with the matching kv heads: the Q/K/V projection rows (columns
for [in,out] weights), the KV-cache columns and the O-projection input slice of those heads.  x
and the RMSNorm are replicated; each rank produces a full-width partial ``o_r[bs, hidden]``; ONE
all-reduce(sum) of bs*hidden fp16 per layer completes the O projection.  This is the contract of
the reference's eager path (fairscale ColumnParallelLinear(wq/wk/wv) + RowParallelLinear(wo),
/root/reference/chat/llama/model.py:208-235); the reference's fused path does not shard at all
(model.py:306-311 gathers the master weights on every rank).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.distributed as dist


@dataclass
class ShardSpec:
    hidden: int
    n_q_heads: int      # global
    n_kv_heads: int     # global
    head_dim: int
    rank: int
    world: int

    def __post_init__(self):
        if self.n_q_heads % self.world or self.n_kv_heads % self.world:
            raise ValueError(f"heads ({self.n_q_heads}q/{self.n_kv_heads}kv) not divisible by world {self.world}")
        if (self.n_q_heads // self.world) % 4:
            raise ValueError("local q heads must be a multiple of 4 (512-wide O-projection strips)")

    @property
    def local_q_heads(self) -> int:
        return self.n_q_heads // self.world

    @property
    def local_kv_heads(self) -> int:
        return self.n_kv_heads // self.world


def shard_layer_weights(weight_qkv: torch.Tensor, weight_o: torch.Tensor, spec: ShardSpec,
                        weight_layout: str = "out_in"):
    """Slice full weights down to this rank's heads (contiguous copies)."""
    hd, D = spec.head_dim, spec.hidden
    qd, kd = spec.n_q_heads * hd, spec.n_kv_heads * hd
    qs = slice(spec.rank * spec.local_q_heads * hd, (spec.rank + 1) * spec.local_q_heads * hd)
    ks = slice(spec.rank * spec.local_kv_heads * hd, (spec.rank + 1) * spec.local_kv_heads * hd)
    if weight_layout == "out_in":
        w = torch.cat([weight_qkv[:qd][qs], weight_qkv[qd:qd + kd][ks], weight_qkv[qd + kd:][ks]], 0).contiguous()
        wo = weight_o[:, qs].contiguous()
    elif weight_layout == "in_out":
        w = torch.cat([weight_qkv[:D][:, qs], weight_qkv[D:2 * D][:, ks], weight_qkv[2 * D:][:, ks]], 0).contiguous()
        wo = weight_o[qs].contiguous()
    else:
        raise ValueError(weight_layout)
    return w, wo


def shard_kv_cache(cache: torch.Tensor, spec: ShardSpec) -> torch.Tensor:
    """[S, n_kv_heads*head_dim] -> this rank's [S, local_kv_heads*head_dim] (contiguous copy)."""
    hd = spec.head_dim
    ks = slice(spec.rank * spec.local_kv_heads * hd, (spec.rank + 1) * spec.local_kv_heads * hd)
    return cache.reshape(cache.shape[0], spec.n_kv_heads * hd)[:, ks].contiguous()


def decoder_layer_tp(local_op: Callable, spec: ShardSpec, group: Optional[dist.ProcessGroup], *args, **kwargs):
    """Run this rank's shard through ``local_op`` (normally clusterfusion_amd.decoder_layer with the
    LOCAL head counts) and complete the O projection with one all-reduce(sum) of the fp16 partial.
    k_new / v_new stay rank-local (this rank's kv heads).  Returns local_op's tuple with ``out``
    replaced by the reduced tensor."""
    res = local_op(*args, n_q_heads=spec.local_q_heads, n_kv_heads=spec.local_kv_heads,
                   head_dim=spec.head_dim, **kwargs)
    out = res[0]
    if spec.world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return (out,) + tuple(res[1:])
