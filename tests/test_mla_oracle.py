"""CPU tests of the DeepSeek MLA oracle (oracle/mla_oracle.py) and of the op's host side.

The reference has NO test, golden vector or eager twin for ``deepseek_decoder_layer``, so this
oracle is "parity unpinned": what can be checked here is that the restated (absorbed) algorithm equals the textbook
un-absorbed MLA formula written independently, that the kernel-rounding emulation stays within fp16 noise of it,
and that the host side of the op validates its arguments without a GPU."""
import ctypes as C

import pytest
import torch

import clusterfusion_amd as cfa
from clusterfusion_amd import _lib
from oracle import mla_oracle as M


@pytest.mark.parametrize("seq_len", [1, 2, 17, 64, 65, 300])
@pytest.mark.parametrize("rope_scores", [False, True])
def test_absorbed_equals_unabsorbed(seq_len, rope_scores):
    inp = M.make_mla_inputs(seq_len + 11, seq_len, score_gain=3.0)
    a = M.mla_decoder_layer(inp, rope_scores=rope_scores)["o"]
    b = M.mla_decoder_layer_unabsorbed(inp, rope_scores=rope_scores)
    assert (a - b).abs().max().item() < 1e-12


def test_last_cache_row_is_never_read():
    """kernel.cuh:470-473: the new token's latent stands in for the last row of the cache."""
    inp = M.make_mla_inputs(3, 40)
    a = M.mla_decoder_layer(inp)["o"]
    inp2 = dict(inp)
    c = inp["ckv_cache"].clone()
    c[-1] = float("nan")
    inp2["ckv_cache"] = c
    b = M.mla_decoder_layer(inp2)["o"]
    assert torch.equal(a, b)


def test_rope_columns_do_not_reach_the_reference_output():
    """kernel.cuh:407-408 reads cache columns 0..511 only, and q_pe / k_pe feed nothing downstream."""
    inp = M.make_mla_inputs(4, 33)
    a = M.mla_decoder_layer(inp)["o"]
    inp2 = dict(inp)
    c = inp["ckv_cache"].clone()
    c[:, 512:] = 7.0
    inp2["ckv_cache"] = c
    inp2["weight_q_pe"] = torch.zeros_like(inp["weight_q_pe"])
    assert torch.equal(a, M.mla_decoder_layer(inp2)["o"])
    assert not torch.equal(M.mla_decoder_layer(inp, rope_scores=True)["o"], M.mla_decoder_layer(inp2, rope_scores=True)["o"])


def test_kernel_rounding_emulation_is_fp16_noise():
    inp = M.make_mla_inputs(5, 128, score_gain=2.0)
    a = M.mla_decoder_layer(inp)["o"]
    e = M.mla_decoder_layer(inp, emulate_kernel_rounding=True)["o"]
    assert (a - e).abs().max().item() < 4e-3 * max(1.0, a.abs().max().item())


def test_rope_half_matches_rotate_half():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(5, 64, generator=g, dtype=torch.float64)
    ang = torch.randn(32, generator=g, dtype=torch.float64)
    cos, sin = torch.cat([ang.cos(), ang.cos()]), torch.cat([ang.sin(), ang.sin()])
    rot = torch.cat([-v[:, 32:], v[:, :32]], dim=-1)
    assert torch.allclose(M.rope_half(v, cos, sin), v * cos + rot * sin, atol=1e-14)


def test_algorithmic_bytes_agree_with_library():
    lib = _lib.load()
    for s in (1, 4096, 65536):
        for r in (0, 1):
            assert lib.cf_deepseek_algorithmic_bytes(s, r) == M.mla_algorithmic_bytes(M.DSV2_LITE, s, bool(r))


def test_c_abi_rejects_bad_arguments():
    lib = _lib.load()
    buf = (C.c_uint16 * 64)()
    p = C.cast(buf, C.c_void_p)
    n = lib.cf_deepseek_workspace_bytes()
    assert n > 8 * 1024 * 1024
    args = lambda **kw: [kw.get("input", p), p, kw.get("q_pe", p), p, p, p, p, p, kw.get("cache", p), kw.get("seq", 16), p, p,
                         p, p, kw.get("eps", 1e-6), kw.get("rope", 0), kw.get("out", p), kw.get("lat", None),
                         kw.get("ws", p), kw.get("wsb", n), None]
    assert lib.cf_deepseek_decoder_layer(*args(input=None)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(out=None)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(seq=0)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(cache=None)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(eps=0.0)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(q_pe=None, rope=1)) == -1
    assert b"rope_scores" in lib.cf_last_error()
    assert lib.cf_deepseek_decoder_layer(*args(ws=None)) == -1
    assert lib.cf_deepseek_decoder_layer(*args(wsb=1024)) == -2


def test_python_op_refuses_cpu_tensors_and_bad_shapes():
    inp = M.make_mla_inputs(0, 8)
    order = ["input", "weight_q_nope", "weight_q_pe", "weight_uk", "weight_kv_nope", "weight_k_pe", "weight_uv", "weight_o",
             "ckv_cache", "rms_input_weight", "rms_ckv_weight", "cos", "sin"]
    with pytest.raises(ValueError, match="GPU"):
        cfa.deepseek_decoder_layer(*[inp[k] for k in order])
    bad = dict(inp)
    bad["input"] = inp["input"].float()
    with pytest.raises(TypeError):
        cfa.deepseek_decoder_layer(*[bad[k] for k in order])
    import clusterfusion
    assert clusterfusion.deepseek_decoder_layer is cfa.deepseek_decoder_layer
