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


def decoder_layer_tp(local_op: Callable, spec: ShardSpec, group: Optional[dist.ProcessGroup], *args,
                     reducer: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, publish_in_kernel: bool = False, **kwargs):
    """Run this rank's shard through ``local_op`` (normally clusterfusion_amd.decoder_layer with the
    LOCAL head counts) and complete the O projection with ONE sum over the ranks of the fp16 partial.
    ``reducer`` is pluggable: ``None`` = ``dist.all_reduce`` over ``group`` (RCCL on ROCm, gloo in the CPU tests);
    a ``OneShotReducer`` = the library's own one-shot all-reduce over peer-mapped buffers (one xGMI link latency for the
    8 KB message of batch 1); any callable ``out -> reduced out`` works.  ``publish_in_kernel=True`` (with a ``OneShotReducer``):
    the shard kernel's phase 3 publishes the partial itself, ``reducer.gather`` completes the sum.  k_new / v_new stay rank-local
    (this rank's kv heads).  Returns local_op's tuple with ``out`` replaced by the reduced tensor."""
    if publish_in_kernel:
        # the publish half of the one-shot all-reduce runs inside the shard's persistent kernel (its phase 3 writes the partial
        # into every rank's receive area); only the gather -- a poll of local memory -- is a launch of its own
        if not isinstance(reducer, OneShotReducer):
            raise ValueError("publish_in_kernel needs reducer=OneShotReducer")
        res = local_op(*args, n_q_heads=spec.local_q_heads, n_kv_heads=spec.local_kv_heads, head_dim=spec.head_dim,
                       tp_publish=reducer, **kwargs)
        return (reducer.gather(res[0].view(-1)).view_as(res[0]),) + tuple(res[1:])
    res = local_op(*args, n_q_heads=spec.local_q_heads, n_kv_heads=spec.local_kv_heads,
                   head_dim=spec.head_dim, **kwargs)
    out = res[0]
    if reducer is not None:
        out = reducer(out)
    elif spec.world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return (out,) + tuple(res[1:])


class _Area:
    """One receive area: fine-grained device memory owned by this process (cf_tp_area_alloc), or a peer's area mapped into it
    (cf_tp_area_import).  `ptr` is the device address in THIS process."""

    def __init__(self, ptr: int, owned: bool, device):
        self.ptr, self.owned, self.device = ptr, owned, device

    @classmethod
    def alloc(cls, nbytes: int, device) -> "_Area":
        import ctypes as C
        from . import _lib
        p = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().cf_tp_area_alloc(nbytes, C.byref(p)))
        return cls(p.value, True, device)

    def export(self) -> bytes:
        import ctypes as C
        from . import _lib
        h = C.create_string_buffer(_lib.CF_TP_HANDLE_BYTES)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().cf_tp_area_export(self.ptr, h))
        return h.raw

    @classmethod
    def from_handle(cls, handle: bytes, device) -> "_Area":
        import ctypes as C
        from . import _lib
        p = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().cf_tp_area_import(C.create_string_buffer(handle, _lib.CF_TP_HANDLE_BYTES), C.byref(p)))
        return cls(p.value, False, device)

    def close(self):
        if self.ptr:
            from . import _lib
            with torch.cuda.device(self.device):
                (_lib.load().cf_tp_area_free if self.owned else _lib.load().cf_tp_area_unmap)(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


import weakref

_live_reducers = weakref.WeakSet()      # clusterfusion_amd.check_device_errors() polls their error words


class OneShotReducer:
    """The one-shot all-reduce of ``cf_tp_oneshot_allreduce`` (include/clusterfusion_hip.h) as a ``decoder_layer_tp`` reducer.

    Every rank owns a receive area; a call writes this rank's fp16 partial into slot ``rank`` of every rank's area (remote
    write-only traffic over xGMI), polls its OWN area until all ``world`` slots carry the call's epoch and sums them in rank
    order in fp32 -- identical bits on every rank, graph-capturable (the epoch lives in the area).  ``areas`` = one entry per rank
    AS MAPPED INTO THIS PROCESS (``areas[rank]`` is this rank's own): ``_Area`` objects (``OneShotReducer.create``: fine-grained
    memory, exchanged as hipIpc handles over the process group -- what ranks on different GPUs need, because peers write while
    the owner's kernel polls) or zeroed 256-byte aligned CUDA uint8 tensors (ranks that share one GPU: the virtual-rank tests).

    Status: the protocol is exercised on ONE GPU (virtual ranks in one process; two processes sharing a device through the
    hipIpc path).  N > 1 over xGMI is unmeasured -- no multi-GPU box was available; ``bench.py --gpus N`` reports RCCL as its
    headline and this reducer (and its in-kernel form: ``gather`` / ``gather_rmsnorm``) as further legs of the same line."""

    def __init__(self, rank: int, world: int, n: int, areas):
        import ctypes as C
        from . import _lib
        if len(areas) != world:
            raise ValueError(f"need {world} areas, got {len(areas)}")
        need = _lib.load().cf_tp_oneshot_bytes(world, n)
        if need == 0:
            raise ValueError(f"unsupported world {world} / n {n}")
        ptrs = []
        for t in areas:
            if isinstance(t, _Area):
                ptrs.append(t.ptr)
                continue
            if t.dtype != torch.uint8 or not t.is_cuda or t.numel() < need or t.data_ptr() % 256:
                raise ValueError(f"every area must be a 256-byte aligned CUDA uint8 tensor of >= {need} bytes")
            ptrs.append(t.data_ptr())
        self.rank, self.world, self.n, self.areas = rank, world, n, list(areas)
        self._ptrs = (C.c_void_p * world)(*ptrs)
        self._lib, self._C = _lib, C
        own = self.areas[rank]
        self.device = own.device
        _live_reducers.add(self)

    @staticmethod
    def area_bytes(world: int, n: int) -> int:
        from . import _lib
        return _lib.load().cf_tp_oneshot_bytes(world, n)

    @classmethod
    def create(cls, group: Optional[dist.ProcessGroup], n: int, device) -> "OneShotReducer":
        """Collective over ``group``: allocate this rank's area, exchange IPC handles, map the peers' areas.
        A step that fails on ONE rank (allocation, an IPC mapping one GPU refuses) fails on EVERY rank: the ranks agree after each
        local step, so nobody is left waiting in a collective for a rank that has already raised."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        own, err = None, None
        try:
            own = _Area.alloc(cls.area_bytes(world, n), device)
            mine = own.export()
        except Exception as e:   # noqa: BLE001 -- reported below, on every rank
            mine, err = None, f"rank {rank}: {type(e).__name__}: {e}"
        handles = [None] * world
        dist.all_gather_object(handles, (mine, err), group=group)
        bad = [h[1] for h in handles if h[0] is None]
        if bad:
            raise RuntimeError("one-shot all-reduce: receive area allocation failed (" + "; ".join(bad) + ")")
        areas = []
        try:
            areas = [own if r == rank else _Area.from_handle(handles[r][0], device) for r in range(world)]
        except Exception as e:   # noqa: BLE001
            err = f"rank {rank}: {type(e).__name__}: {e}"
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)     # (also the barrier: every rank has mapped every area before the first call writes into them)
        bad = [e for e in errs if e]
        if bad:
            raise RuntimeError("one-shot all-reduce: mapping a peer's receive area failed (" + "; ".join(bad) + ")")
        return cls(rank, world, n, areas)

    def __call__(self, out: torch.Tensor, *, publish_only: bool = False) -> torch.Tensor:
        if out.dtype != torch.float16 or not out.is_cuda or out.numel() != self.n or not out.is_contiguous():
            raise ValueError(f"expected a contiguous CUDA fp16 tensor of {self.n} elements")
        with torch.cuda.device(out.device):
            self._lib.check(self._lib.load().cf_tp_oneshot_allreduce(
                out.data_ptr(), out.data_ptr(), self.n, self.rank, self.world, self._ptrs, int(publish_only),
                torch.cuda.current_stream(out.device).cuda_stream))
        return out

    # ---- the publish folded into the layer kernel (prepare_decoder_layer(..., tp_publish=reducer)) + the gather half ----------
    def gather(self, out: torch.Tensor) -> torch.Tensor:
        """The gather half alone (``cf_tp_gather``): ``out`` := the fp32 sum in rank order, rounded to fp16, of the partials the
        ranks' layer kernels published in their phase 3.  Same bits as ``__call__`` on the partials, one launch that only polls
        local memory; advances the area's epoch.  A peer that never publishes: NaN in ``out``, error word 7, sticky word raised."""
        if out.dtype != torch.float16 or not out.is_cuda or out.numel() != self.n or not out.is_contiguous():
            raise ValueError(f"expected a contiguous CUDA fp16 tensor of {self.n} elements")
        with torch.cuda.device(out.device):
            self._lib.check(self._lib.load().cf_tp_gather(out.data_ptr(), self.n, self.rank, self.world, self._ptrs,
                                                          torch.cuda.current_stream(out.device).cuda_stream))
        return out

    def gather_rmsnorm(self, weight: torch.Tensor, eps: float, *, residual=None, residual_out=None, out=None, sum_out=None):
        """The gather folded into the fused add + RMSNorm that follows the attention block (``cf_rmsnorm_tp_gather``):
        sum = all-reduced attention output (-> ``sum_out``), h = sum + residual (-> ``residual_out``), returns
        RMSNorm(h) * weight.  One launch instead of gather + norm."""
        dev = weight.device
        if out is None:
            out = torch.empty(1, self.n, dtype=torch.float16, device=dev)
        for t in (weight, residual, residual_out, out, sum_out):
            if t is not None and (t.dtype != torch.float16 or not t.is_cuda or t.numel() != self.n or not t.is_contiguous()):
                raise ValueError(f"expected contiguous CUDA fp16 tensors of {self.n} elements")
        p = lambda t: None if t is None else t.data_ptr()     # noqa: E731
        with torch.cuda.device(dev):
            self._lib.check(self._lib.load().cf_rmsnorm_tp_gather(self._ptrs, self.rank, self.world, p(residual), p(weight), float(eps), self.n,
                                                                  p(out), p(residual_out), p(sum_out),
                                                                  torch.cuda.current_stream(dev).cuda_stream))
        return out

    def clear_error(self) -> None:
        own = self.areas[self.rank]
        if not isinstance(own, _Area):
            own[4:8].zero_()
            return
        with torch.cuda.device(own.device):
            self._lib.check(self._lib.load().cf_tp_area_clear_error(own.ptr, torch.cuda.current_stream(own.device).cuda_stream))

    def error(self) -> int:
        """Word 1 of this rank's area (0 = fine, 7 = a peer's slot never arrived); synchronises the current stream."""
        own = self.areas[self.rank]
        if not isinstance(own, _Area):
            return int(own[4:8].view(torch.int32).item())
        code = self._C.c_uint32()
        with torch.cuda.device(own.device):
            self._lib.check(self._lib.load().cf_tp_area_status(own.ptr, torch.cuda.current_stream(own.device).cuda_stream, self._C.byref(code)))
        return int(code.value)
