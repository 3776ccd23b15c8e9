"""Python operator API of the fused Llama decoder-layer decode path on MI355X.

Mirrors, name for name and argument for argument, the pybind module of the reference
(/root/reference/include/pybind.cpp:108-116, re-exported by clusterfusion/__init__.py:6-16):

    llama_decoder_layer(input, weight_qkv, weight_o, k_cache, v_cache, rms_input_weight, cos, sin)
        -> (o, k, v)                                                  pybind.cpp:3-12,110
    llama_decoder_layer_sglang(input, residual, weight_qkv, weight_o, k_cache, v_cache,
        rms_input_weight, eps, cos, sin) -> (o, residual, k, v)        pybind.cpp:14-25,111
    llama_decoder_layer_batch_decode_sglang(output, residual_output, input, residual, weight_qkv,
        weight_o, paged_kv_indptr, paged_kv_indices, k_cache_ptrs, v_cache_ptrs, layer_id,
        rms_input_weight, eps, positions, cos_sin) -> None             pybind.cpp:27-43,112

plus ``decoder_layer`` -- the superset the three are thin views of (run-time dims for GQA and
head-parallel TP shards, KV page size > 1).

PyTorch is plumbing here (device memory, the current HIP stream); all arithmetic happens in
hand-written HIP behind the C-ABI (include/clusterfusion_hip.h).  Differences from the reference
that are deliberate: launches go to torch's CURRENT stream with no device
synchronisation (reference: legacy stream between two cudaDeviceSynchronize,
llama_kernel_dispatch.cu:126,144); the device is the input's, not cuda:0 (:18); shapes / dtypes /
devices are validated (reference: none) and errors raise.
"""
from __future__ import annotations

import collections
import ctypes as C
import warnings
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import cf_dims, cf_layer_args

__all__ = [
    "llama_decoder_layer", "llama_decoder_layer_sglang", "llama_decoder_layer_batch_decode_sglang",
    "decoder_layer", "prepare_decoder_layer", "PreparedLayer", "workspace_bytes", "algorithmic_bytes", "profile_enable", "profile_read",
    "host_binding", "set_host_binding", "host_binding_stats", "set_tuning", "set_path", "last_path", "last_variant", "last_arm", "check_device_errors", "rmsnorm", "set_weight_relayout", "invalidate_weight_relayout", "release_weight_relayout", "weight_relayout_stats",
    "deepseek_decoder_layer", "deepseek_algorithmic_bytes", "deepseek_profile",
]

_HIDDEN, _HEADS, _HEAD_DIM = 4096, 32, 128       # reference config.h:2-11 (Llama-2-7B)
_GQA_KV_HEADS = 8                                # ... and the grouped-query sibling the sglang-style entries also take (Llama-3-8B / Mistral-7B: 32 q / 8 kv)
_workspaces = {}

# ---- the compiled host binding (csrc/cf_torch_binding.cpp) ----------------------------------------------------------------------
# The reference's entries are direct C++ functions behind pybind11 (include/pybind.cpp:108-112) and its caller issues one eager
# call per layer per token (chat/llama/model.py:358-367): a layer is 27-35 us of GPU time, so the host side of a call has to be
# a few microseconds, not the ~60 of fifteen Python-level checks, three torch.empty and a 20-argument ctypes call.  The three
# Llama entries therefore go through ``_cf_fast`` first: same checks, output allocation and C-ABI call in C++.  It takes a call
# only when nothing needs setting up or reporting and answers NotImplemented otherwise; then the Python body below runs -- it
# owns every error message and the workspace / re-layout caches (and registers them with the binding).  CF_NO_FAST=1 switches
# the binding off (A/B of the two host paths; both end in the same C entry points of the same loaded library).
_fast = None
_fast_state = {"tried": False}


def _fast_binding():
    """The compiled binding module, or None (CF_NO_FAST=1, or it is not built: warned once -- the ctypes path is complete)."""
    global _fast
    if _fast_state["tried"]:
        return _fast
    _fast_state["tried"] = True
    import os
    if os.environ.get("CF_NO_FAST") == "1":
        return None
    lib = _lib.load()
    try:
        from . import _cf_fast as m
    except ImportError as e:
        warnings.warn(f"clusterfusion_amd: the compiled host binding _cf_fast is not available ({e}); every call takes the ctypes "
                      "path (~10x the host time per call). Build it with `python -m clusterfusion_amd.build`", RuntimeWarning)
        return None

    def addr(name):
        return C.cast(getattr(lib, name), C.c_void_p).value
    m.bind(addr("cf_llama_decoder_layer"), addr("cf_llama_decoder_layer_out_in"), addr("cf_llama_decoder_layer_sglang"),
           addr("cf_llama_decoder_layer_batch_decode_sglang"), addr("cf_last_error"), addr("cf_decoder_layer_ex"), _lib.CFError)
    _fast = m
    return m


def host_binding() -> str:
    """"compiled" when the three Llama entries run through the C++ binding (_cf_fast), "ctypes" otherwise."""
    return "compiled" if _fast_binding() is not None else "ctypes"


def set_host_binding(kind: str) -> None:
    """"compiled" / "ctypes": which host path the three Llama entries take from now on (same-process A/B; both end in the same C
    entry points).  "compiled" raises if the binding is not built."""
    global _fast
    if kind == "ctypes":
        _fast_state["tried"], _fast = True, None
        return
    if kind != "compiled":
        raise ValueError(f"host binding {kind!r}")
    _fast_state["tried"] = False
    import os
    was = os.environ.pop("CF_NO_FAST", None)
    try:
        if _fast_binding() is None:
            raise RuntimeError("the compiled host binding is not built (python -m clusterfusion_amd.build)")
    finally:
        if was is not None:
            os.environ["CF_NO_FAST"] = was
    # what the Python side set up while the binding was off
    _fast.ws_clear()
    for (dev_index, stream, hidden, hq, hkv, batch), ws in _workspaces.items():
        if (hidden, hq) == (_HIDDEN, _HEADS) and hkv in (_HEADS, _GQA_KV_HEADS):
            _fast.ws_register(dev_index, stream, batch, hkv, ws.data_ptr(), ws.numel())
    _fast_sync_relayout()


def host_binding_stats() -> dict:
    """{"taken", "declined", "workspaces", "weight_copies"} of the compiled binding (zeros when it is off)."""
    if _fast is None:
        return {"taken": 0, "declined": 0, "workspaces": 0, "weight_copies": 0}
    t, d, w, r = _fast.stats()
    return {"taken": t, "declined": d, "workspaces": w, "weight_copies": r}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t, name, dtype, device=None, numel=None, min_numel=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got {t.device}); there is no CPU path")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got shape {tuple(t.shape)}")
    if min_numel is not None and t.numel() < min_numel:
        raise ValueError(f"{name}: expected at least {min_numel} elements, got shape {tuple(t.shape)}")
    return t


def _dev(device) -> torch.device:
    """Normalised device: 'cuda' -> cuda:<current>."""
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


def _workspace(dims: cf_dims, batch: int, device: torch.device) -> torch.Tensor:
    lib = _lib.load()
    device = _dev(device)
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream, dims.hidden, dims.n_q_heads, dims.n_kv_heads, batch)
    ws = _workspaces.get(key)
    if ws is None:
        n = lib.cf_workspace_bytes(C.byref(dims), batch)
        if n == 0:
            raise _lib.CFError("cf_workspace_bytes returned 0 (unsupported dims)")
        # set up ONCE (cf_workspace_init: zeroed + failure-report address): it carries the persistent kernel's epoch
        # counter and tagged granules
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(lib.cf_workspace_init(ws.data_ptr(), n, stream.cuda_stream))
        _workspaces[key] = ws
        if (dims.hidden, dims.n_q_heads, dims.head_dim) == (_HIDDEN, _HEADS, _HEAD_DIM) and dims.n_kv_heads in (_HEADS, _GQA_KV_HEADS):
            f = _fast_binding()
            if f is not None:
                f.ws_register(device.index, stream.cuda_stream, batch, dims.n_kv_heads, ws.data_ptr(), n)
    return ws


def workspace_bytes(hidden=_HIDDEN, n_q_heads=_HEADS, n_kv_heads=_HEADS, head_dim=_HEAD_DIM, batch=1) -> int:
    return _lib.load().cf_workspace_bytes(C.byref(cf_dims(hidden, n_q_heads, n_kv_heads, head_dim)), batch)


def algorithmic_bytes(seq_len, hidden=_HIDDEN, n_q_heads=_HEADS, n_kv_heads=_HEADS, head_dim=_HEAD_DIM,
                      batch=1, residual=True) -> int:
    d = cf_dims(hidden, n_q_heads, n_kv_heads, head_dim)
    return _lib.load().cf_algorithmic_bytes(C.byref(d), batch, seq_len, int(residual))


def profile_enable(on: bool = True) -> None:
    _lib.check(_lib.load().cf_profile_enable(int(on)))


def profile_read(reset: bool = True):
    """-> (per-stage milliseconds [qkv, attention, o-proj, reduce], number of calls)."""
    ms = (C.c_double * _lib.CF_PROFILE_STAGES)()
    n = C.c_int64(0)
    _lib.check(_lib.load().cf_profile_read(ms, C.byref(n), int(reset)))
    return list(ms), n.value


def set_tuning(kv_splits: int = 0) -> None:
    _lib.check(_lib.load().cf_set_tuning(kv_splits))


def set_path(path: str = "auto") -> None:
    """"auto" (fused persistent kernel when the shape qualifies), "pipeline", or "fused" (required)."""
    _lib.check(_lib.load().cf_set_path({"auto": 0, "pipeline": 1, "fused": 2}[path]))


def last_path() -> str:
    """Which path the last layer call of this thread took: "pipeline", "fused" or "none"."""
    return {0: "none", 1: "pipeline", 2: "fused"}[_lib.load().cf_last_path()]


def last_variant() -> str:
    """Which kernel the last layer call of this thread ran, e.g. "k_fused_decode_mha<IO=false>", "k_fused_decode_g<8, 4>"
    or "stage pipeline".  The length arm of a persistent kernel is chosen on the device: ``last_arm``."""
    return _lib.load().cf_last_variant().decode()


ARMS = {0: "none", 1: "two tiles", 2: "one 128-token tile", 3: "one 256-token tile", 4: "tile loop"}


def last_arm(n_q_heads=_HEADS, n_kv_heads=None, batch=1, hidden=_HIDDEN, device=None) -> str:
    """Which length arm the persistent kernel took in the LAST completed call on the current stream's workspace for these
    dims (synchronises the stream): the kernels read the cached length on the device and branch there, so one captured
    graph serves a growing sequence.  One of ARMS' values."""
    device = _dev(device if device is not None else torch.device("cuda"))
    n_kv_heads = n_q_heads if n_kv_heads is None else n_kv_heads
    stream = torch.cuda.current_stream(device)
    ws = _workspaces.get((device.index, stream.cuda_stream, hidden, n_q_heads, n_kv_heads, batch))
    if ws is None:
        return ARMS[0]
    arm = C.c_uint32(0)
    with torch.cuda.device(device):
        _lib.check(_lib.load().cf_workspace_last_arm(ws.data_ptr(), stream.cuda_stream, C.byref(arm)))
    return ARMS.get(arm.value, f"arm {arm.value}")


def check_device_errors(device=None) -> None:
    """Synchronise the stream every cached workspace belongs to and raise if a persistent kernel reported a failed
    inter-workgroup exchange there (its 256 workgroups were not co-resident: something else was using the GPU).  The
    workspace is set up afresh before raising, so the next call works; the outputs of the failed call are invalid.
    Without this poll the failure still surfaces at the next layer CALL on the device (C-ABI: CF_ELAUNCH) -- but a loop that
    only replays a captured graph makes no calls: poll this (e.g. once per generated token, next to the sampling sync)."""
    lib = _lib.load()
    want = None if device is None else _dev(device)
    failed = []
    for key, ws in list(_workspaces.items()) + list(_mla_workspaces.items()):
        if want is not None and torch.device("cuda", key[0]) != want:
            continue
        code = C.c_uint32(0)
        with torch.cuda.device(ws.device):
            # key[1] = the raw handle of the stream the workspace was created on and is used by
            _lib.check(lib.cf_workspace_status(ws.data_ptr(), key[1], C.byref(code)))
            if code.value:
                failed.append((key, code.value))
                _lib.check(lib.cf_workspace_init(ws.data_ptr(), ws.numel(), key[1]))
    from . import tp as _tp
    tp_failed = []
    for red in list(_tp._live_reducers):
        if want is not None and _dev(red.device) != want:
            continue
        code = red.error()
        if code:
            tp_failed.append((red, code))
            red.clear_error()
    tp_text = ""
    if tp_failed:
        for dev_index in sorted({_dev(r.device).index for r, _ in tp_failed}):
            with torch.cuda.device(torch.device("cuda", dev_index)):
                lib.cf_take_sticky_error()
        tp_text = ("TP gather(s) timed out: " + ", ".join(f"rank {r.rank}/{r.world} code {c}" for r, c in tp_failed) +
                   " (a peer never published its partial: the reduced outputs of those calls are NaN); error words cleared")
    if failed:
        # the kernels also raised the per-device sticky words: consume them here (every failed device), this exception is the report
        for dev_index in sorted({k[0] for k, _ in failed}):
            with torch.cuda.device(torch.device("cuda", dev_index)):
                lib.cf_take_sticky_error()
        # both kinds in one poll: one exception that names both (the TP failure used to be cleared unreported)
        raise _lib.CFError("persistent kernel exchange(s) timed out: " +
                           ", ".join(f"device {k[0]} code {c}" for k, c in failed) +
                           " (workgroups not co-resident: another stream or process held CUs); the workspace was re-initialised" +
                           ("; ALSO " + tp_text if tp_text else ""))
    if tp_text:
        raise _lib.CFError(tp_text)


class PreparedLayer:
    """A validated, reusable call: the argument block is built once, ``run()`` only stamps the
    current stream and crosses the C-ABI (keeps the per-call host cost at one ctypes call, which
    matters because a layer is ~35 us of GPU time).  Holds references to every tensor it points at."""

    __slots__ = ("args", "device", "outputs", "_keep", "_stream")

    def __init__(self, args, device, outputs, keep, stream):
        self.args, self.device, self.outputs, self._keep, self._stream = args, device, outputs, keep, stream

    def with_tp_publish(self, reducer) -> "PreparedLayer":
        """The same call (same tensors, same outputs) whose kernel ALSO publishes its partial output into the receive areas of
        ``reducer`` (a ``clusterfusion_amd.tp.OneShotReducer``): see ``prepare_decoder_layer(tp_publish=)``."""
        if self.args.batch != 1 or self.args.dims.hidden != reducer.n:
            raise ValueError("with_tp_publish: batch 1 and hidden == the reducer's n expected")
        a = type(self.args).from_buffer_copy(self.args)
        a.tp_areas, a.tp_rank, a.tp_world = reducer._ptrs, reducer.rank, reducer.world
        return PreparedLayer(a, self.device, self.outputs, self._keep + [reducer], self._stream)

    def run(self):
        """Launch on torch's current stream of the layer's device.  The exchange workspace belongs to ONE stream (two
        streams must never run the persistent kernel on one workspace concurrently): when the current stream is not the
        one the call was prepared on, the workspace of the current stream is used instead (set up at that moment: switch
        streams OUTSIDE a capture first -- cf_workspace_init's memset and copy do not belong in a graph)."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._stream:
            with torch.cuda.device(self.device):
                ws = _workspace(self.args.dims, self.args.batch, self.device)
            # (the module-level cache owns the per-stream workspaces: nothing to keep here, alternating streams grows nothing)
            self.args.workspace, self.args.workspace_bytes = ws.data_ptr(), ws.numel()
            self._stream = cur
        self.args.stream = cur
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                rc = _lib.load().cf_decoder_layer_ex(C.byref(self.args))
        else:
            rc = _lib.load().cf_decoder_layer_ex(C.byref(self.args))
        if rc:
            _lib.check(rc)
        return self.outputs


def decoder_layer(*args, **kwargs):
    """Superset entry (C-ABI ``cf_decoder_layer_ex``): ``prepare_decoder_layer(...).run()``.
    Returns (out, residual_out, k_new, v_new)."""
    p = prepare_decoder_layer(*args, **kwargs)
    with torch.cuda.device(p.device):
        return p.run()


def prepare_decoder_layer(
    x: torch.Tensor, residual: Optional[torch.Tensor], weight_qkv: torch.Tensor, weight_o: torch.Tensor,
    k_cache: Optional[torch.Tensor], v_cache: Optional[torch.Tensor], rms_weight: torch.Tensor, eps: float,
    cos: torch.Tensor, sin: torch.Tensor, *,
    n_q_heads: int = _HEADS, n_kv_heads: Optional[int] = None, head_dim: int = _HEAD_DIM,
    weight_layout: str = "out_in", rope_style: str = "neox",
    kv_indptr: Optional[torch.Tensor] = None, kv_indices: Optional[torch.Tensor] = None,
    kv_seq_lens: Optional[torch.Tensor] = None, page_size: int = 1, max_seq_len: int = 0,
    kv_cache_ptrs: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, layer_id: int = 0,
    positions: Optional[torch.Tensor] = None, rope_row_stride: int = 0,
    out: Optional[torch.Tensor] = None, residual_out: Optional[torch.Tensor] = None,
    k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None,
    write_kv_to_cache: bool = False, want_kv: bool = True, tp_publish=None,
):
    """Validate once and build the C argument block; see ``decoder_layer``.

    ``tp_publish``: a ``clusterfusion_amd.tp.OneShotReducer`` -- phase 3 of the shard's persistent kernel then ALSO writes its
    partial output into every rank's receive area (the publish half of the one-shot all-reduce, no launch of its own); the call
    must be followed on the same stream by ``reducer.gather(out)`` or ``reducer.gather_rmsnorm(...)``.

    x [batch, hidden] fp16.  Contiguous KV mode (kv_indptr None): batch 1, k_cache/v_cache
    [S, n_kv_heads*128].  Paged mode: k_cache/v_cache are slot arrays [num_slots, n_kv_heads*128]
    (or ``kv_cache_ptrs`` = (uint64[n_layers], uint64[n_layers]) device pointer tables + layer_id),
    kv_indptr int32 [batch+1], kv_indices int32 (token slots for page_size 1, page ids otherwise),
    kv_seq_lens int32 [batch] (required for page_size > 1).
    For a head-parallel TP shard pass the LOCAL head counts and weight shards; the caller
    all-reduces ``out`` (see clusterfusion_amd.tp).
    """
    lib = _lib.load()
    n_kv_heads = n_q_heads if n_kv_heads is None else n_kv_heads
    if weight_layout not in ("out_in", "in_out"):
        raise ValueError(f"weight_layout {weight_layout!r}")
    if rope_style not in ("neox", "gptj"):
        raise ValueError(f"rope_style {rope_style!r}")
    x = _need(x, "x", torch.float16)
    dev = _dev(x.device)
    rms_weight = _need(rms_weight, "rms_weight", torch.float16, dev)
    hidden = rms_weight.numel()
    if hidden == 0 or x.numel() % hidden:
        raise ValueError(f"x: {tuple(x.shape)} is not a whole number of rows of hidden={hidden}")
    batch = x.numel() // hidden
    q_dim, kv_dim = n_q_heads * head_dim, n_kv_heads * head_dim
    wq_rows = (q_dim + 2 * kv_dim) if weight_layout == "out_in" else 3 * hidden
    wq_cols = hidden if weight_layout == "out_in" else q_dim
    weight_qkv = _need(weight_qkv, "weight_qkv", torch.float16, dev, numel=wq_rows * wq_cols)
    weight_o = _need(weight_o, "weight_o", torch.float16, dev, numel=hidden * q_dim)
    if residual is not None:
        residual = _need(residual, "residual", torch.float16, dev, numel=batch * hidden)
    n_ang = head_dim // 2 if rope_style == "neox" else head_dim
    cos = _need(cos, "cos", torch.float32, dev, min_numel=n_ang)
    sin = _need(sin, "sin", torch.float32, dev, min_numel=n_ang)

    a = cf_layer_args()
    a.dims = cf_dims(hidden, n_q_heads, n_kv_heads, head_dim)
    a.batch = batch
    a.weight_layout = _lib.CF_W_OUT_IN if weight_layout == "out_in" else _lib.CF_W_IN_OUT
    a.rope_style = _lib.CF_ROPE_NEOX if rope_style == "neox" else _lib.CF_ROPE_GPTJ
    a.eps = float(eps)
    a.x, a.residual = _ptr(x), _ptr(residual)
    a.weight_qkv, a.weight_o, a.rms_weight = _ptr(weight_qkv), _ptr(weight_o), _ptr(rms_weight)
    a.page_size = int(page_size)
    if kv_indptr is None:
        if batch != 1:
            raise ValueError("contiguous KV mode is single-sequence: x must be one row")
        if k_cache is None or v_cache is None:
            raise ValueError("k_cache / v_cache required")
        k_cache = _need(k_cache, "k_cache", torch.float16, dev)
        v_cache = _need(v_cache, "v_cache", torch.float16, dev)
        if k_cache.numel() % kv_dim or k_cache.numel() != v_cache.numel():
            raise ValueError(f"k_cache/v_cache: expected [S, {kv_dim}] each, got {tuple(k_cache.shape)} / {tuple(v_cache.shape)}")
        a.seq_len = k_cache.numel() // kv_dim
        a.k_cache, a.v_cache = _ptr(k_cache), _ptr(v_cache)
    else:
        kv_indptr = _need(kv_indptr, "kv_indptr", torch.int32, dev, numel=batch + 1)
        kv_indices = _need(kv_indices, "kv_indices", torch.int32, dev)
        a.kv_indptr, a.kv_indices = _ptr(kv_indptr), _ptr(kv_indices)
        if kv_seq_lens is not None:
            kv_seq_lens = _need(kv_seq_lens, "kv_seq_lens", torch.int32, dev, numel=batch)
            a.kv_seq_lens = _ptr(kv_seq_lens)
        elif page_size != 1:
            raise ValueError("page_size > 1 needs kv_seq_lens")
        if kv_cache_ptrs is not None:
            kp = _need(kv_cache_ptrs[0], "k_cache_ptrs", torch.uint64, dev) if kv_cache_ptrs[0].dtype == torch.uint64 \
                else _need(kv_cache_ptrs[0], "k_cache_ptrs", torch.int64, dev)
            vp = _need(kv_cache_ptrs[1], "v_cache_ptrs", kp.dtype, dev)
            if not (0 <= layer_id < kp.numel()):
                raise ValueError(f"layer_id {layer_id} outside k_cache_ptrs[{kp.numel()}]")
            a.kv_cache_ptrs_k, a.kv_cache_ptrs_v, a.layer_id = _ptr(kp), _ptr(vp), int(layer_id)
        else:
            k_cache = _need(k_cache, "k_cache", torch.float16, dev)
            v_cache = _need(v_cache, "v_cache", torch.float16, dev)
            a.k_cache, a.v_cache = _ptr(k_cache), _ptr(v_cache)
        a.max_seq_len = int(max_seq_len)
        a.write_kv_to_cache = int(write_kv_to_cache)
    if positions is not None:
        positions = _need(positions, "positions", torch.int64, dev, numel=batch)
        a.positions, a.rope_row_stride = _ptr(positions), int(rope_row_stride)
    a.cos, a.sin = _ptr(cos), _ptr(sin)

    if out is None:
        out = torch.empty(batch, hidden, dtype=torch.float16, device=dev)
    else:
        _need(out, "out", torch.float16, dev, numel=batch * hidden)
    if residual is not None and residual_out is None:
        residual_out = torch.empty(batch, hidden, dtype=torch.float16, device=dev)
    if residual_out is not None:
        _need(residual_out, "residual_out", torch.float16, dev, numel=batch * hidden)
    if want_kv and k_new is None:
        k_new = torch.empty(batch, n_kv_heads, head_dim, dtype=torch.float16, device=dev)
    if want_kv and v_new is None:
        v_new = torch.empty(batch, n_kv_heads, head_dim, dtype=torch.float16, device=dev)
    for t, nm in ((k_new, "k_new"), (v_new, "v_new")):
        if t is not None:
            _need(t, nm, torch.float16, dev, numel=batch * kv_dim)
    a.out, a.residual_out, a.k_new, a.v_new = _ptr(out), _ptr(residual_out), _ptr(k_new), _ptr(v_new)
    if tp_publish is not None:
        if batch != 1 or hidden != tp_publish.n:
            raise ValueError(f"tp_publish: batch 1 and hidden == the reducer's n ({tp_publish.n}) expected, got batch {batch}, hidden {hidden}")
        a.tp_areas, a.tp_rank, a.tp_world = tp_publish._ptrs, tp_publish.rank, tp_publish.world

    with torch.cuda.device(dev):
        ws = _workspace(a.dims, batch, dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    keep = [x, residual, weight_qkv, weight_o, rms_weight, k_cache, v_cache, kv_indptr, kv_indices, kv_seq_lens,
            kv_cache_ptrs, positions, cos, sin, ws, tp_publish]
    return PreparedLayer(a, dev, (out, residual_out, k_new, v_new), keep, torch.cuda.current_stream(dev).cuda_stream)


def _llama2_checks(input, weight_qkv, weight_o, rms_input_weight, gqa_ok=False):
    """-> device (and, with ``gqa_ok``, the number of kv heads weight_qkv's size implies: 32, or 8 for a [6144, 4096] matrix)."""
    _need(input, "input", torch.float16, numel=None)
    dev = input.device
    hkv = _HEADS
    if gqa_ok and isinstance(weight_qkv, torch.Tensor) and weight_qkv.numel() == (_HEADS + 2 * _GQA_KV_HEADS) * _HEAD_DIM * _HIDDEN:
        hkv = _GQA_KV_HEADS
        _need(weight_qkv, "weight_qkv", torch.float16, dev)
    else:
        _need(weight_qkv, "weight_qkv", torch.float16, dev, numel=3 * _HIDDEN * _HIDDEN)
    _need(weight_o, "weight_o", torch.float16, dev, numel=_HIDDEN * _HIDDEN)
    _need(rms_input_weight, "rms_input_weight", torch.float16, dev, numel=_HIDDEN)
    return (dev, hkv) if gqa_ok else dev


# The plain entry is served from weights re-laid out ONCE to [out,in] -- the orientation whose phase 1 streams whole 8-KB
# rows; the reference's [in,out] (chat/llama/model.py:317-322) makes every head read a strided 256-B piece per input row
# (29 vs 32.5 us per layer at S=1024, 35 vs 40 at S=4096).  The trade, stated at the boundary:
#   * it costs a second copy of a layer's weights (134 MB for Llama-2-7B, 4.3 GB for its 32 layers) and one transpose at the
#     layer's first call; the FIRST population warns once (``ResourceWarning``) with these numbers;
#   * POINTERS ARE STABLE: a copy lives at one address from its first call until it is released, whatever happens to the
#     weights in between -- a captured hipGraph or a ``PreparedLayer`` holds raw pointers to it.  A changed version counter or
#     ``invalidate_weight_relayout()`` re-lays out IN PLACE into the same buffers (on the current stream: replays issued
#     afterwards see the new weights);
#   * an entry PINS the tensors it was made from (strong references: the address of a dead original could otherwise be reused
#     by a new tensor and hit a stale copy; a per-call transient -- ``w.data``, ``w.detach()`` -- shares that memory and hits
#     the entry).  A caller that unloads a model releases the copies AND the pins with
#     ``release_weight_relayout()`` / ``set_weight_relayout(False)``;
#   * the copies are capped by a byte budget (default 16 GiB of the GPU's 288); over budget the least recently used copy that
#     no capture has seen is released; entries used while a stream was capturing are never evicted (a graph points at them);
#     when nothing can go, the layer runs the native [in,out] kernel;
#   * RELEASING frees memory a captured graph may still point at: after ``release_weight_relayout()`` or
#     ``set_weight_relayout(False)`` every graph captured through this entry must be captured again;
#   * staleness contract: writes the version counters cannot see -- ``param.data.copy_()``, raw-pointer or RCCL writes into the
#     same storage -- MUST be followed by ``invalidate_weight_relayout()`` (all entries) or ``invalidate_weight_relayout(w)``;
#   * nothing is allocated or transposed during stream capture: a first call inside a capture runs the native [in,out] kernel
#     (and warns) -- warm the layer up outside the capture, as torch.cuda.graphs users do anyway;
#   * the C-ABI entry ``cf_llama_decoder_layer`` never re-lays anything out (a C caller uses ``cf_relayout_weights`` once and
#     ``cf_llama_decoder_layer_out_in``).
_relayout = {"on": True, "cache": collections.OrderedDict(), "bytes": 0, "budget": 16 << 30, "warned": False, "warned_capture": False}


def _fast_sync_relayout() -> None:
    """Hand the compiled binding the current set of copies (it serves a plain-entry call only from a registered, current copy;
    every change of the cache -- a new copy, a refresh, a release, an eviction -- goes through here)."""
    f = _fast_binding()
    if f is None:
        return
    f.relayout_clear()
    for ent in _relayout["cache"].values():
        if ent["ver"] is not None:
            f.relayout_register(ent["src"][0], ent["src"][1], ent["seen"][0], ent["seen"][1], ent["wq"], ent["wo"])


def set_weight_relayout(on: bool = True, max_bytes: Optional[int] = None) -> None:
    """Switch the [in,out] -> [out,in] weight cache of ``llama_decoder_layer`` (default: on, 16 GiB budget); turning it off
    releases every copy (graphs captured through the entry must then be captured again).  ``max_bytes`` changes the budget."""
    _relayout["on"] = bool(on)
    if max_bytes is not None:
        _relayout["budget"] = int(max_bytes)
    if not on:
        release_weight_relayout()


def _matching(weight):
    return [k for k in _relayout["cache"] if weight is None or weight.data_ptr() in k]


def invalidate_weight_relayout(weight: Optional[torch.Tensor] = None) -> None:
    """Re-lay out, IN PLACE and now (current stream), the copy made from ``weight`` (a weight_qkv or weight_o tensor the plain
    entry was called with), or every copy when None.  REQUIRED after updating weights through a path the tensors' version
    counters cannot see (``param.data.copy_()``, raw-pointer / RCCL writes).  The copies keep their addresses: captured graphs
    stay valid and see the new weights in every replay issued after this call."""
    keys = _matching(weight)
    if keys and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("invalidate_weight_relayout() during stream capture: the re-layout would be captured into the graph")
    for key in keys:
        _relay(_relayout["cache"][key])
    _fast_sync_relayout()


def release_weight_relayout(weight: Optional[torch.Tensor] = None) -> None:
    """Free the copy made from ``weight`` (or every copy) together with the pin on its source tensors.  Any hipGraph or
    ``PreparedLayer`` captured through ``llama_decoder_layer`` with those weights points at freed memory afterwards: capture
    again."""
    for key in _matching(weight):
        _relayout["bytes"] -= _relayout["cache"].pop(key)["bytes"]
    _fast_sync_relayout()


def weight_relayout_stats() -> dict:
    """{"entries", "bytes", "budget", "pinned_by_capture"} of the re-laid-out weight copies currently held."""
    c = _relayout["cache"]
    return {"entries": len(c), "bytes": _relayout["bytes"], "budget": _relayout["budget"],
            "pinned_by_capture": sum(1 for e in c.values() if e["captured"])}


def _ver(t) -> int:
    """The tensor's version counter; inference tensors (made under ``torch.inference_mode()``) have none and read as 0 -- their
    in-place updates are then invisible here: ``invalidate_weight_relayout()`` after changing such weights."""
    return 0 if t.is_inference() else t._version


def _relay(ent) -> None:
    """(re-)fill an entry's buffers from its pinned sources with the library's own transpose (cf_relayout_weights)"""
    wq_src, wo_src = ent["src"]
    dev = _dev(wq_src.device)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().cf_relayout_weights(C.byref(cf_dims(_HIDDEN, _HEADS, _HEADS, _HEAD_DIM)), wq_src.data_ptr(),
                                                   wo_src.data_ptr(), ent["wq"].data_ptr(), ent["wo"].data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream))
    ent["ver"] = (_ver(wq_src), _ver(wo_src))


def _relaid_out(weight_qkv, weight_o):
    """-> (wq [12288, 4096], wo [4096, 4096]) in [out,in] orientation, or None (no room in the budget, or a first call
    during stream capture): the caller then runs the native [in,out] kernel."""
    cache = _relayout["cache"]
    key = (weight_qkv.data_ptr(), weight_o.data_ptr())
    capturing = torch.cuda.is_current_stream_capturing()
    hit = cache.get(key)
    if hit is not None:
        # Stale when the pinned sources' version counters moved -- or the PASSED tensors' did: an alias carries a counter of its
        # own (first call with the transient `w.data`, later calls with `w` after in-place updates: only w's counter sees them;
        # ADVICE r4).  Pass the SAME tensor objects every call: a caller that alternates aliases with different counters (`w` and
        # `w.data`) pays a re-layout per switch -- never a stale copy -- and one that only ever passes `w.data` must call
        # invalidate_weight_relayout() after updating the weights (no counter it shows ever moves).
        seen = (_ver(weight_qkv), _ver(weight_o))
        src_moved = hit["ver"] != (_ver(hit["src"][0]), _ver(hit["src"][1]))
        if src_moved or hit["seen"] != seen:
            if capturing:
                # nothing may be transposed inside a capture.  The pinned sources moved: the copy IS stale.  Only the alias's
                # counter differs: the weights themselves did not change as far as any counter shows -- keep the copy
                if src_moved:
                    raise RuntimeError("llama_decoder_layer: the weights changed since their re-laid-out copy was made and the stream is "
                                       "capturing; call the layer (or invalidate_weight_relayout) once outside the capture")
            else:
                _relay(hit)                                     # modified in place: same buffers, same addresses
                hit["seen"] = seen
                _fast_sync_relayout()
        hit["captured"] |= capturing
        cache.move_to_end(key)
        return hit["wq"], hit["wo"]
    if capturing:
        if not _relayout["warned_capture"]:
            _relayout["warned_capture"] = True
            warnings.warn("clusterfusion_amd.llama_decoder_layer: first call for these weights happens during stream capture -- "
                          "nothing is allocated or transposed inside a capture, the native [in,out] kernel is captured instead "
                          "(about 3.5 us per layer slower); call the layer once outside the capture first", RuntimeWarning, stacklevel=3)
        return None
    need = (weight_qkv.numel() + weight_o.numel()) * 2
    if need > _relayout["budget"]:
        return None
    while _relayout["bytes"] + need > _relayout["budget"]:      # the least recently used copy no graph has seen goes
        victim = next((k for k, e in cache.items() if not e["captured"]), None)
        if victim is None:
            return None
        _relayout["bytes"] -= cache.pop(victim)["bytes"]
        _fast_sync_relayout()
    if not _relayout["warned"]:
        _relayout["warned"] = True
        warnings.warn(f"clusterfusion_amd.llama_decoder_layer keeps a re-laid-out [out,in] copy of each layer's weights "
                      f"({need >> 20} MiB per layer, budget {_relayout['budget'] >> 30} GiB, LRU) for its faster kernel and pins the "
                      "originals while it does; after weight updates the version counters cannot see (param.data.copy_, raw "
                      "pointers) call invalidate_weight_relayout(); release_weight_relayout() / set_weight_relayout(False) frees "
                      "copies and pins", ResourceWarning, stacklevel=3)
    ent = {"wq": torch.empty_like(weight_qkv), "wo": torch.empty_like(weight_o), "bytes": need, "src": (weight_qkv, weight_o),
           "ver": None, "seen": (_ver(weight_qkv), _ver(weight_o)), "captured": False}
    _relay(ent)
    cache[key] = ent
    _relayout["bytes"] += need
    _fast_sync_relayout()
    return ent["wq"], ent["wo"]


def llama_decoder_layer(input, weight_qkv, weight_o, k_cache, v_cache, rms_input_weight, cos, sin):
    """Drop-in for ``clusterfusion.llama_decoder_layer`` (pybind.cpp:110; call site
    chat/llama/model.py:358-367).  Llama-2-7B, [in,out] weights ([12288,4096] / [4096,4096]),
    k_cache/v_cache [S,4096] post-RoPE, cos/sin fp32 [1,128] pair-duplicated (GPT-J), eps 1e-6.
    Returns (o [1,4096] -- no residual add, k [1,32,128] post-RoPE, v [1,32,128])."""
    if _fast is not None or (not _fast_state["tried"] and _fast_binding() is not None):
        try:
            r = _fast.llama_decoder_layer(input, weight_qkv, weight_o, k_cache, v_cache, rms_input_weight, cos, sin)
        except TypeError:           # (an argument that is no tensor at all: the checks below say which)
            r = NotImplemented
        if r is not NotImplemented:
            return r
    lib = _lib.load()
    dev = _llama2_checks(input, weight_qkv, weight_o, rms_input_weight)
    if input.numel() != _HIDDEN:
        raise ValueError(f"input: expected 4096 elements (one token), got {tuple(input.shape)}")
    relaid = _relaid_out(weight_qkv, weight_o) if _relayout["on"] else None
    if relaid is not None:
        wq, wo = relaid
        o, _, k, v = decoder_layer(input.reshape(1, _HIDDEN), None, wq, wo, k_cache.reshape(-1, _HIDDEN),
                                   v_cache.reshape(-1, _HIDDEN), rms_input_weight, 1e-6, cos, sin,
                                   weight_layout="out_in", rope_style="gptj")
        return o, k.view(1, _HEADS, _HEAD_DIM), v.view(1, _HEADS, _HEAD_DIM)
    k_cache = _need(k_cache, "k_cache", torch.float16, dev)
    v_cache = _need(v_cache, "v_cache", torch.float16, dev)
    if k_cache.numel() % _HIDDEN or k_cache.numel() != v_cache.numel():
        raise ValueError("k_cache/v_cache: expected [S, 4096] each")
    cos = _need(cos, "cos", torch.float32, dev, min_numel=_HEAD_DIM)
    sin = _need(sin, "sin", torch.float32, dev, min_numel=_HEAD_DIM)
    o = torch.empty(1, _HIDDEN, dtype=torch.float16, device=dev)
    k = torch.empty(1, _HEADS, _HEAD_DIM, dtype=torch.float16, device=dev)
    v = torch.empty(1, _HEADS, _HEAD_DIM, dtype=torch.float16, device=dev)
    ws = _workspace(cf_dims(_HIDDEN, _HEADS, _HEADS, _HEAD_DIM), 1, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.cf_llama_decoder_layer(
            input.data_ptr(), weight_qkv.data_ptr(), weight_o.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
            k_cache.numel() // _HIDDEN, rms_input_weight.data_ptr(), cos.data_ptr(), sin.data_ptr(),
            o.data_ptr(), k.data_ptr(), v.data_ptr(), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(dev).cuda_stream))
    return o, k, v


def rmsnorm(input, weight, eps: float = 1e-6, *, residual=None, residual_out=None, out=None):
    """Drop-in for ``clusterfusion.rmsnorm(input, weight)`` (pybind.cpp:113; tests/test_norm.py: input [64, 8192]):
    ``fp16(x * rsqrt(mean(x^2) + eps) * weight)`` per row.  With ``residual`` it is the fused add + RMSNorm
    (x := input + residual); fp16(x) goes to ``residual_out`` when given (may be ``residual`` itself)."""
    lib = _lib.load()
    input = _need(input, "input", torch.float16)
    dev = input.device
    hidden = input.shape[-1]
    rows = input.numel() // hidden
    weight = _need(weight, "weight", torch.float16, dev, numel=hidden)
    if residual is not None:
        residual = _need(residual, "residual", torch.float16, dev, numel=input.numel())
    if residual_out is not None:
        if residual is None:
            raise ValueError("residual_out without residual")
        residual_out = _need(residual_out, "residual_out", torch.float16, dev, numel=input.numel())
    out = torch.empty_like(input) if out is None else _need(out, "out", torch.float16, dev, numel=input.numel())
    with torch.cuda.device(dev):
        _lib.check(lib.cf_rmsnorm(input.data_ptr(), residual.data_ptr() if residual is not None else None,
                                  weight.data_ptr(), float(eps), rows, hidden, out.data_ptr(),
                                  residual_out.data_ptr() if residual_out is not None else None,
                                  torch.cuda.current_stream(dev).cuda_stream))
    return out


# DeepSeek-V2-Lite MLA dims (include/H100/deepseek/config.h:2-9)
_MLA_HIDDEN, _MLA_HEADS, _MLA_NOPE, _MLA_ROPE, _MLA_LORA = 2048, 16, 128, 64, 512
_MLA_LATENT = _MLA_LORA + _MLA_ROPE
_mla_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def _mla_workspace(device: torch.device) -> torch.Tensor:
    device = _dev(device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _mla_workspaces.get(key)
    if ws is None:
        # set up ONCE (cf_workspace_init): it carries the epoch counter and tagged granules of the hand-offs
        n = _lib.load().cf_deepseek_workspace_bytes()
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.load().cf_workspace_init(ws.data_ptr(), n, key[1]))
        _mla_workspaces[key] = ws
    return ws


def deepseek_decoder_layer(input, weight_q_nope, weight_q_pe, weight_uk, weight_kv_nope, weight_k_pe, weight_uv,
                           weight_o, ckv_cache, rms_input_weight, rms_ckv_weight, cos, sin, *,
                           eps: float = 1e-6, rope_scores: bool = False, return_latent: bool = False):
    """Drop-in for ``clusterfusion.deepseek_decoder_layer`` (pybind.cpp:45-59,113; deepseek_kernel_dispatch.cu:4-242):
    the MLA attention block of DeepSeek-V2-Lite for ONE new token.  All weights [in,out] fp16:
    weight_q_nope [2048, 2048], weight_q_pe [2048, 1024], weight_uk [128, 8192], weight_kv_nope [2048, 512],
    weight_k_pe [2048, 64], weight_uv [512, 2048], weight_o [2048, 2048]; cos / sin fp32 [64];
    ckv_cache [S, 576] whose LAST row is the new token's slot (not read; the reference fixes S = 4096).  -> o [1, 2048].

    Keyword extensions (defaults = the reference's behaviour): ``rope_scores`` adds RoPE(q_pe) . k_pe to the scores
    (the reference's kernel computes both vectors and never uses them, kernel.cuh:298-315,407-408);
    ``return_latent`` also returns the [576] row (RMSNorm(ckv) | RoPE(k_pe)) a caller appends to the cache."""
    lib = _lib.load()
    input = _need(input, "input", torch.float16, numel=_MLA_HIDDEN)
    dev = input.device
    H, N, R, L, D = _MLA_HEADS, _MLA_NOPE, _MLA_ROPE, _MLA_LORA, _MLA_HIDDEN
    weight_q_nope = _need(weight_q_nope, "weight_q_nope", torch.float16, dev, numel=D * H * N)
    weight_q_pe = _need(weight_q_pe, "weight_q_pe", torch.float16, dev, numel=D * H * R)
    weight_uk = _need(weight_uk, "weight_uk", torch.float16, dev, numel=N * H * L)
    weight_kv_nope = _need(weight_kv_nope, "weight_kv_nope", torch.float16, dev, numel=D * L)
    weight_k_pe = _need(weight_k_pe, "weight_k_pe", torch.float16, dev, numel=D * R)
    weight_uv = _need(weight_uv, "weight_uv", torch.float16, dev, numel=L * H * N)
    weight_o = _need(weight_o, "weight_o", torch.float16, dev, numel=H * N * D)
    ckv_cache = _need(ckv_cache, "ckv_cache", torch.float16, dev)
    if ckv_cache.numel() % _MLA_LATENT or ckv_cache.numel() == 0:
        raise ValueError(f"ckv_cache: expected [S >= 1, {_MLA_LATENT}], got {tuple(ckv_cache.shape)}")
    rms_input_weight = _need(rms_input_weight, "rms_input_weight", torch.float16, dev, numel=D)
    rms_ckv_weight = _need(rms_ckv_weight, "rms_ckv_weight", torch.float16, dev, numel=L)
    cos = _need(cos, "cos", torch.float32, dev, min_numel=R)
    sin = _need(sin, "sin", torch.float32, dev, min_numel=R)
    o = torch.empty(1, D, dtype=torch.float16, device=dev)
    latent = torch.empty(_MLA_LATENT, dtype=torch.float16, device=dev) if return_latent else None
    ws = _mla_workspace(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.cf_deepseek_decoder_layer(
            input.data_ptr(), weight_q_nope.data_ptr(), weight_q_pe.data_ptr(), weight_uk.data_ptr(),
            weight_kv_nope.data_ptr(), weight_k_pe.data_ptr(), weight_uv.data_ptr(), weight_o.data_ptr(),
            ckv_cache.data_ptr(), ckv_cache.numel() // _MLA_LATENT, rms_input_weight.data_ptr(),
            rms_ckv_weight.data_ptr(), cos.data_ptr(), sin.data_ptr(), float(eps), int(bool(rope_scores)),
            o.data_ptr(), latent.data_ptr() if latent is not None else None, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(dev).cuda_stream))
    return (o, latent) if return_latent else o


def deepseek_algorithmic_bytes(seq_len: int, rope_scores: bool = False) -> int:
    return _lib.load().cf_deepseek_algorithmic_bytes(int(seq_len), int(bool(rope_scores)))


def deepseek_profile(on: Optional[bool] = None, reset: bool = True):
    """on=True/False: switch the per-stage hipEvent timing of deepseek_decoder_layer (synchronises every call).
    on=None: read -> (stage_ms[3] accumulated: projections + absorbed query (or the whole persistent launch) | attention |
    W_uv + W_o, calls)."""
    lib = _lib.load()
    if on is not None:
        _lib.check(lib.cf_deepseek_profile_enable(int(on)))
        return None
    ms = (C.c_double * _lib.CF_MLA_STAGES)()
    n = C.c_int64(0)
    _lib.check(lib.cf_deepseek_profile_read(ms, C.byref(n), int(reset)))
    return list(ms), n.value


def llama_decoder_layer_sglang(input, residual, weight_qkv, weight_o, k_cache, v_cache, rms_input_weight,
                               eps, cos, sin):
    """Drop-in for ``clusterfusion.llama_decoder_layer_sglang`` (pybind.cpp:111;
    tests/test_llama.py:145-156).  [out,in] weights, NEOX RoPE (first 64 of cos/sin), ``residual``
    is updated IN PLACE to fp16(input + residual) and returned.  -> (o, residual, k, v).
    Extension: weight_qkv [6144, 4096] selects the grouped-query geometry (32 q / 8 kv heads, Llama-3-8B / Mistral-7B: BASELINE
    config 4) -- caches [S, 1024], k / v [1, 8, 128]."""
    if _fast is not None or (not _fast_state["tried"] and _fast_binding() is not None):
        try:
            r = _fast.llama_decoder_layer_sglang(input, residual, weight_qkv, weight_o, k_cache, v_cache, rms_input_weight, eps, cos, sin)
        except TypeError:
            r = NotImplemented
        if r is not NotImplemented:
            return r
    lib = _lib.load()
    dev, hkv = _llama2_checks(input, weight_qkv, weight_o, rms_input_weight, gqa_ok=True)
    if input.numel() != _HIDDEN:
        raise ValueError(f"input: expected 4096 elements (one token), got {tuple(input.shape)}")
    residual = _need(residual, "residual", torch.float16, dev, numel=_HIDDEN)
    k_cache = _need(k_cache, "k_cache", torch.float16, dev)
    v_cache = _need(v_cache, "v_cache", torch.float16, dev)
    if hkv != _HEADS:
        # Extension: weight_qkv [6144, 4096] = the grouped-query geometry (32 q / 8 kv heads; caches [S, 1024]).  The reference's kernels are
        # compiled for 32 / 32 (config.h:2-11); its eager model defines the grouping (repeat_kv, chat/llama/model.py:166-175).
        kvd = hkv * _HEAD_DIM
        if k_cache.numel() % kvd or k_cache.numel() != v_cache.numel():
            raise ValueError(f"k_cache/v_cache: expected [S, {kvd}] each for {hkv} kv heads")
        o, _, k, v = decoder_layer(input.reshape(1, _HIDDEN), residual, weight_qkv, weight_o, k_cache.reshape(-1, kvd), v_cache.reshape(-1, kvd),
                                   rms_input_weight, eps, cos, sin, n_q_heads=_HEADS, n_kv_heads=hkv, residual_out=residual)
        return o, residual, k, v
    if k_cache.numel() % _HIDDEN or k_cache.numel() != v_cache.numel():
        raise ValueError("k_cache/v_cache: expected [S, 4096] each")
    cos = _need(cos, "cos", torch.float32, dev, min_numel=_HEAD_DIM // 2)
    sin = _need(sin, "sin", torch.float32, dev, min_numel=_HEAD_DIM // 2)
    o = torch.empty(1, _HIDDEN, dtype=torch.float16, device=dev)
    k = torch.empty(1, _HEADS, _HEAD_DIM, dtype=torch.float16, device=dev)
    v = torch.empty(1, _HEADS, _HEAD_DIM, dtype=torch.float16, device=dev)
    ws = _workspace(cf_dims(_HIDDEN, _HEADS, _HEADS, _HEAD_DIM), 1, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.cf_llama_decoder_layer_sglang(
            input.data_ptr(), residual.data_ptr(), weight_qkv.data_ptr(), weight_o.data_ptr(),
            k_cache.data_ptr(), v_cache.data_ptr(), k_cache.numel() // _HIDDEN, rms_input_weight.data_ptr(),
            float(eps), cos.data_ptr(), sin.data_ptr(), o.data_ptr(), k.data_ptr(), v.data_ptr(),
            ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream))
    return o, residual, k, v


def llama_decoder_layer_batch_decode_sglang(output, residual_output, input, residual, weight_qkv, weight_o,
                                            paged_kv_indptr, paged_kv_indices, k_cache_ptrs, v_cache_ptrs,
                                            layer_id, rms_input_weight, eps, positions, cos_sin):
    """Drop-in for ``clusterfusion.llama_decoder_layer_batch_decode_sglang`` (pybind.cpp:112).
    Writes ``output``/``residual_output`` [bs,4096] and the new token's K/V into cache slot
    ``paged_kv_indices[paged_kv_indptr[b+1]-1]`` of the layer's caches.  Returns None.
    Extension: weight_qkv [6144, 4096] selects the grouped-query geometry (32 q / 8 kv heads; per-layer caches [num_slots, 1024])."""
    if _fast is not None or (not _fast_state["tried"] and _fast_binding() is not None):
        try:
            if _fast.llama_decoder_layer_batch_decode_sglang(output, residual_output, input, residual, weight_qkv, weight_o, paged_kv_indptr,
                                                             paged_kv_indices, k_cache_ptrs, v_cache_ptrs, layer_id, rms_input_weight, eps,
                                                             positions, cos_sin) is None:
                return None
        except TypeError:
            pass
    lib = _lib.load()
    dev, hkv = _llama2_checks(input, weight_qkv, weight_o, rms_input_weight, gqa_ok=True)
    if input.numel() % _HIDDEN:
        raise ValueError(f"input: expected [bs, 4096], got {tuple(input.shape)}")
    bs = input.numel() // _HIDDEN
    _need(output, "output", torch.float16, dev, numel=bs * _HIDDEN)
    _need(residual_output, "residual_output", torch.float16, dev, numel=bs * _HIDDEN)
    _need(residual, "residual", torch.float16, dev, numel=bs * _HIDDEN)
    _need(paged_kv_indptr, "paged_kv_indptr", torch.int32, dev, numel=bs + 1)
    _need(paged_kv_indices, "paged_kv_indices", torch.int32, dev)
    pdt = k_cache_ptrs.dtype if isinstance(k_cache_ptrs, torch.Tensor) else None
    if pdt not in (torch.uint64, torch.int64):
        raise TypeError("k_cache_ptrs / v_cache_ptrs: expected uint64 (or int64) device-pointer tensors")
    _need(k_cache_ptrs, "k_cache_ptrs", pdt, dev)
    _need(v_cache_ptrs, "v_cache_ptrs", pdt, dev, numel=k_cache_ptrs.numel())
    if not (0 <= int(layer_id) < k_cache_ptrs.numel()):
        raise ValueError(f"layer_id {layer_id} outside k_cache_ptrs[{k_cache_ptrs.numel()}]")
    _need(positions, "positions", torch.int64, dev, numel=bs)
    _need(cos_sin, "cos_sin", torch.float32, dev, min_numel=_HEAD_DIM)
    if hkv != _HEADS:      # the grouped-query geometry (see llama_decoder_layer_sglang): caches [num_slots, 1024]
        decoder_layer(input.reshape(bs, _HIDDEN), residual.reshape(bs, _HIDDEN), weight_qkv, weight_o, None, None, rms_input_weight, eps,
                      cos_sin, cos_sin.view(-1)[_HEAD_DIM // 2:], n_q_heads=_HEADS, n_kv_heads=hkv, kv_indptr=paged_kv_indptr,
                      kv_indices=paged_kv_indices, kv_cache_ptrs=(k_cache_ptrs, v_cache_ptrs), layer_id=int(layer_id), positions=positions,
                      rope_row_stride=_HEAD_DIM, out=output.reshape(bs, _HIDDEN), residual_out=residual_output.reshape(bs, _HIDDEN),
                      want_kv=False, write_kv_to_cache=True, max_seq_len=max(int(paged_kv_indices.numel()) - bs, 1) if bs <= 4 else 0)
        return None
    ws = _workspace(cf_dims(_HIDDEN, _HEADS, _HEADS, _HEAD_DIM), bs, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.cf_llama_decoder_layer_batch_decode_sglang(
            output.data_ptr(), residual_output.data_ptr(), input.data_ptr(), residual.data_ptr(),
            weight_qkv.data_ptr(), weight_o.data_ptr(), paged_kv_indptr.data_ptr(), paged_kv_indices.data_ptr(),
            k_cache_ptrs.data_ptr(), v_cache_ptrs.data_ptr(), int(layer_id), rms_input_weight.data_ptr(),
            float(eps), positions.data_ptr(), cos_sin.data_ptr(), bs,
            # planning bound of any row's cached length, known to the host without a sync: the index array's size
            # (lets 1 .. 4 sequences reach the persistent kernels, as a prepared call does).  For larger batches the
            # sum over all rows says nothing about one row: "unknown", or the split planner over-splits (batch 16: +16 us)
            max(int(paged_kv_indices.numel()) - bs, 1) if bs <= 4 else 0, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(dev).cuda_stream))
    return None
