"""CPU: the C-ABI library loads, exports every symbol include/clusterfusion_hip.h declares, and
rejects bad arguments before touching a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest
import torch

import clusterfusion_amd as cfa
from clusterfusion_amd import _lib
from clusterfusion_amd import build as cfbuild
from oracle import cf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    cfbuild.build()
    return _lib.load()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "clusterfusion_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_all_exported(lib):
    names = _declared_symbols()
    assert "cf_llama_decoder_layer" in names and "cf_decoder_layer_ex" in names
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.EXPORTS), "ctypes binding table out of sync with the header"


def test_struct_layout_matches_c(lib):
    # cf_layer_args is 8-byte aligned; spot-check the size the C side was compiled with by a
    # round trip through an error path that reads the LAST field (workspace_bytes / stream unused)
    assert C.sizeof(_lib.cf_dims) == 16
    assert C.sizeof(_lib.cf_layer_args) % 8 == 0


def test_workspace_and_bytes(lib):
    assert cfa.workspace_bytes() > 0
    assert cfa.workspace_bytes(batch=4) > cfa.workspace_bytes(batch=1)
    assert cfa.workspace_bytes(head_dim=64) >= 0
    for dims, S in ((O.LLAMA2_7B, 4096), (O.LLAMA2_7B, 128), (O.LLAMA3_8B, 8192)):
        got = cfa.algorithmic_bytes(S, dims.hidden, dims.n_q_heads, dims.n_kv_heads, dims.head_dim)
        assert got == O.algorithmic_bytes(dims, S)


def test_c_abi_rejects_bad_arguments(lib):
    assert lib.cf_decoder_layer_ex(None) == -1
    assert b"NULL" in lib.cf_last_error()
    a = _lib.cf_layer_args()
    a.dims = _lib.cf_dims(4096, 32, 32, 64)
    a.batch = 1
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -4          # head_dim 64 unsupported
    a.dims = _lib.cf_dims(4096, 32, 32, 128)
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -1          # NULL pointers
    a.dims = _lib.cf_dims(4096, 32, 5, 128)
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -1          # 32 % 5
    # misaligned tensor pointer (every kernel uses 16-byte vector loads)
    a.dims = _lib.cf_dims(4096, 32, 32, 128)
    buf = (C.c_uint8 * 256)()
    base = C.addressof(buf)
    base += (-base) % 16
    for f in ("x", "weight_qkv", "weight_o", "rms_weight", "cos", "sin", "out"):
        setattr(a, f, base)
    a.weight_qkv = base + 2
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -1 and b"aligned" in lib.cf_last_error()
    assert lib.cf_set_tuning(1000) == -1 and lib.cf_set_tuning(0) == 0


def test_python_ops_fail_loudly_without_gpu_tensors(lib):
    inp = O.make_inputs(0, 4, O.LayerDims(1024, 8, 8, 128))
    with pytest.raises(ValueError, match="no CPU path"):
        cfa.decoder_layer(inp["x"], None, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                          inp["rms_w"], 1e-6, inp["cos"], inp["sin"], n_q_heads=8)
    with pytest.raises(TypeError):
        cfa.llama_decoder_layer(inp["x"].float(), None, None, None, None, None, None, None)
    with pytest.raises(TypeError):
        cfa.llama_decoder_layer("x", None, None, None, None, None, None, None)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.load()


def test_reference_import_names():
    import clusterfusion
    for n in ("llama_decoder_layer", "llama_decoder_layer_sglang", "llama_decoder_layer_batch_decode_sglang"):
        assert callable(getattr(clusterfusion, n))


def test_rmsnorm_c_abi_rejects_bad_arguments(lib):
    """cf_rmsnorm validates before launching anything (no GPU here)."""
    buf = (C.c_uint16 * 64)()
    p = C.cast(buf, C.c_void_p)
    assert lib.cf_rmsnorm(None, None, p, 1e-6, 1, 64, p, None, None) == -1        # NULL input
    assert lib.cf_rmsnorm(p, None, p, 1e-6, 0, 64, p, None, None) == -1           # no rows
    assert lib.cf_rmsnorm(p, None, p, 1e-6, 1, 60, p, None, None) == -4           # hidden not a multiple of 8
    assert lib.cf_rmsnorm(p, None, p, 1e-6, 1, 16384, p, None, None) == -4        # wider than one workgroup covers
    assert lib.cf_rmsnorm(p, None, p, 1e-6, 1, 64, p, p, None) == -1              # residual_out without residual
    assert b"residual" in lib.cf_last_error()


def test_python_ops_refuse_cpu_tensors_and_bad_shapes():
    """The drop-in entries have no CPU path: CPU tensors / wrong dtypes / wrong shapes raise before any launch."""
    x = torch.zeros(4, 512, dtype=torch.float16)
    w = torch.ones(512, dtype=torch.float16)
    with pytest.raises(ValueError, match="GPU"):
        cfa.rmsnorm(x, w)
    with pytest.raises(TypeError):
        cfa.rmsnorm(x.float(), w)
    with pytest.raises(TypeError):
        cfa.rmsnorm([1, 2, 3], w)
    out = torch.zeros(2, 4096, dtype=torch.float16)
    with pytest.raises((ValueError, TypeError)):
        cfa.llama_decoder_layer_batch_decode_sglang(out, out, out, out, torch.zeros(12288, 4096, dtype=torch.float16),
                                                    torch.zeros(4096, 4096, dtype=torch.float16),
                                                    torch.zeros(3, dtype=torch.int32), torch.zeros(8, dtype=torch.int32),
                                                    torch.zeros(1, dtype=torch.uint64), torch.zeros(1, dtype=torch.uint64), 0,
                                                    torch.ones(4096, dtype=torch.float16), 1e-6,
                                                    torch.zeros(2, dtype=torch.int64), torch.zeros(16, 128))


def test_weight_relayout_switch_is_cheap_without_gpu():
    cfa.set_weight_relayout(False)
    cfa.set_weight_relayout(True, max_bytes=1 << 30)
    cfa.set_weight_relayout(True, max_bytes=16 << 30)      # the default


def test_small_host_entries_without_gpu(lib):
    """cf_workspace_bytes validates the dims before it sizes anything (zeroed dims used to divide by zero);
    cf_last_variant is a static string; the debug hook validates its arguments before launching."""
    d = _lib.cf_dims(0, 0, 0, 0)
    assert lib.cf_workspace_bytes(C.byref(d), 1) == 0
    d = _lib.cf_dims(4096, 32, 0, 128)
    assert lib.cf_workspace_bytes(C.byref(d), 1) == 0
    d = _lib.cf_dims(4096, 32, 32, 128)
    assert lib.cf_workspace_bytes(C.byref(d), 1) > 256 and lib.cf_workspace_bytes(C.byref(d), 0) == 0
    assert isinstance(cfa.last_variant(), str)
    assert lib.cf_debug_occupy(None, 0, 0, 0) == -1 and lib.cf_debug_occupy(None, 1, 1 << 20, 0) == -1
    assert lib.cf_workspace_init(None, 0, None) == -1
    # cf_relayout_weights: argument checks come before any launch
    d = _lib.cf_dims(4096, 32, 32, 128)
    assert lib.cf_relayout_weights(C.byref(d), None, 8, 16, 24, None) == -1            # null source
    assert lib.cf_relayout_weights(C.byref(d), 8, 16, 8, 24, None) == -1               # in place
    g = _lib.cf_dims(4096, 32, 8, 128)
    assert lib.cf_relayout_weights(C.byref(g), 8, 16, 24, 32, None) == -4              # grouped-query: no [in,out] orientation


def test_harness_and_shim_import_without_gpu():
    import clusterfusion
    from clusterfusion_amd import harness
    assert set(clusterfusion.__all__) == {"llama_decoder_layer", "llama_decoder_layer_sglang",
                                          "llama_decoder_layer_batch_decode_sglang", "rmsnorm",
                                          "deepseek_decoder_layer"}
    cos, sin = harness.precompute_rotary(128, 16)
    assert cos.shape == (16, 128) and torch.allclose(cos[:, 0], cos[:, 1]) and torch.allclose(cos[0], torch.ones(128))
    assert hasattr(harness, "DecodeModel") and hasattr(harness, "FusedAttentionBlock")


def test_out_in_entry_and_last_arm_reject_bad_arguments(lib):
    """cf_llama_decoder_layer_out_in (the documented way for a C / pybind caller to reach the fast kernel with weights it
    re-laid out once) validates like the plain entry; cf_workspace_last_arm checks its pointers before any device access."""
    buf = (C.c_uint8 * 256)()
    base = C.addressof(buf)
    base += (-base) % 16
    args = [base] * 5 + [4] + [base] * 6 + [base, 0, None]          # workspace too small
    assert lib.cf_llama_decoder_layer_out_in(*args) == -2 and b"workspace" in lib.cf_last_error()
    args[0] = None
    assert lib.cf_llama_decoder_layer_out_in(*args) == -1           # NULL input
    args[0], args[1] = base, base + 2
    assert lib.cf_llama_decoder_layer_out_in(*args) == -1 and b"aligned" in lib.cf_last_error()
    arm = C.c_uint32(7)
    assert lib.cf_workspace_last_arm(None, None, C.byref(arm)) == -1 and arm.value == 7
    assert lib.cf_workspace_last_arm(base, None, None) == -1


def test_tp_oneshot_rejects_bad_arguments(lib):
    """The one-shot all-reduce of head-parallel TP validates rank / world / n / area pointers before launching."""
    assert lib.cf_tp_oneshot_bytes(8, 4096) == (32 + 2 * 8 * 2048) * 8 and lib.cf_tp_oneshot_bytes(9, 4096) == 0
    assert lib.cf_tp_oneshot_bytes(2, 4095) == 0
    buf = (C.c_uint8 * 1024)()
    base = C.addressof(buf)
    base += (-base) % 256
    areas = (C.c_void_p * 2)(base, base)
    assert lib.cf_tp_oneshot_allreduce(None, base, 4096, 0, 2, areas, 0, None) == -1
    assert lib.cf_tp_oneshot_allreduce(base, base, 4096, 2, 2, areas, 0, None) == -1 and b"rank" in lib.cf_last_error()
    assert lib.cf_tp_oneshot_allreduce(base, base, 4095, 0, 2, areas, 0, None) == -1
    areas[1] = base + 8
    assert lib.cf_tp_oneshot_allreduce(base, base, 4096, 0, 2, areas, 0, None) == -1 and b"aligned" in lib.cf_last_error()


def test_tp_area_helpers_reject_bad_arguments(lib):
    """The receive-area helpers (fine-grained allocation, hipIpc export / import) validate before they touch HIP; freeing or
    unmapping nothing is fine."""
    p = C.c_void_p()
    code = C.c_uint32()
    handle = (C.c_uint8 * 64)()
    assert lib.cf_tp_area_alloc(0, C.byref(p)) == -1 and lib.cf_tp_area_alloc(4096, None) == -1
    assert lib.cf_tp_area_export(None, handle) == -1 and lib.cf_tp_area_export(C.addressof(handle), None) == -1
    assert lib.cf_tp_area_import(None, C.byref(p)) == -1 and lib.cf_tp_area_import(handle, None) == -1
    assert lib.cf_tp_area_status(None, None, C.byref(code)) == -1 and lib.cf_tp_area_status(C.addressof(handle), None, None) == -1
    assert lib.cf_tp_area_free(None) == 0 and lib.cf_tp_area_unmap(None) == 0


def test_tp_gather_entry_points_reject_bad_arguments(lib):
    """ABI 2: the gather half of the TP collective (`cf_tp_gather`), its fused add + RMSNorm form (`cf_rmsnorm_tp_gather`) and the
    `tp_areas` fields of `cf_layer_args` validate before anything is launched."""
    assert lib.cf_abi_version() == 2
    buf = (C.c_uint8 * 1024)()
    base = C.addressof(buf)
    base += (-base) % 256
    areas = (C.c_void_p * 2)(base, base)
    assert lib.cf_tp_gather(None, 4096, 0, 2, areas, None) == -1
    assert lib.cf_tp_gather(base, 4096, 0, 2, None, None) == -1
    assert lib.cf_tp_gather(base, 4096, 2, 2, areas, None) == -1 and b"rank" in lib.cf_last_error()
    assert lib.cf_tp_gather(base, 4095, 0, 2, areas, None) == -1
    assert lib.cf_rmsnorm_tp_gather(areas, 0, 2, None, None, 1e-6, 4096, base, None, None, None) == -1          # no weight
    assert lib.cf_rmsnorm_tp_gather(areas, 0, 2, None, base, 1e-6, 4000, base, None, None, None) == -4          # hidden not a multiple of 512
    assert lib.cf_rmsnorm_tp_gather(areas, 0, 2, None, base, 1e-6, 4096, base, base, None, None) == -1          # residual_out without residual
    assert lib.cf_rmsnorm_tp_gather(areas, 0, 9, None, base, 1e-6, 4096, base, None, None, None) == -1 and b"world" in lib.cf_last_error()
    areas[1] = base + 8
    assert lib.cf_tp_gather(base, 4096, 0, 2, areas, None) == -1 and b"aligned" in lib.cf_last_error()
    assert lib.cf_tp_area_clear_error(None, None) == -1
    # cf_layer_args.tp_*: rank / world / alignment are checked with the other arguments (no GPU needed to get that far)
    a = _lib.cf_layer_args()
    a.dims = _lib.cf_dims(4096, 4, 4, 128)
    a.batch, a.weight_layout, a.rope_style, a.eps = 1, _lib.CF_W_OUT_IN, _lib.CF_ROPE_NEOX, 1e-6
    for f in ("x", "weight_qkv", "weight_o", "rms_weight", "cos", "sin", "out", "k_cache", "v_cache"):
        setattr(a, f, base)
    a.seq_len = 1
    good = (C.c_void_p * 2)(base, base)
    a.tp_areas, a.tp_rank, a.tp_world = good, 2, 2
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -1 and b"tp_rank" in lib.cf_last_error()
    a.tp_areas, a.tp_rank = areas, 0
    assert lib.cf_decoder_layer_ex(C.byref(a)) == -1 and b"tp_areas" in lib.cf_last_error()
