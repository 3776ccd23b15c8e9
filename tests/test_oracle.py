"""CPU: pin the oracle (oracle/cf_oracle.py) against fixtures minted from the reference's
own Python (oracle/gen_golden.py), and check its internal consistency."""
import math

import numpy as np
import pytest
import torch

from oracle import cf_oracle as O
from tests._util import GOLDEN, golden_inputs, load_golden, max_abs, max_err_in_ulps_of_max, max_ulp, ulp16

NEOX = ["neox_s1_tl", "neox_s37_tl", "neox_s256_tl", "neox_s128", "neox_s1024"]
GPTJ = ["gptj_s64", "gptj_s1024"]


def test_helpers_match_reference_model_py():
    z = np.load(f"{GOLDEN}/helpers.npz")
    t = torch.from_numpy(z["rope_in"])
    cos = torch.repeat_interleave(torch.from_numpy(z["rope_cos"]), 2, dim=-1)
    sin = torch.repeat_interleave(torch.from_numpy(z["rope_sin"]), 2, dim=-1)
    got = O.rope(t, cos, sin, "gptj")
    assert max_abs(got, torch.from_numpy(z["rope_out"])) < 1e-6
    y = O.rms_norm(torch.from_numpy(z["rms_in"]), torch.from_numpy(z["rms_w"]), 1e-6)
    assert max_abs(y, torch.from_numpy(z["rms_out"])) < 1e-5


@pytest.mark.parametrize("name", NEOX)
def test_oracle_matches_reference_eager_neox(name):
    meta, gold = load_golden(name)
    dims, inp = golden_inputs(meta)
    assert O.input_checksum(inp) == meta["input_sha256"], "RNG drift: regenerate goldens"
    out, res, k, v = O.decoder_layer(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"],
                                     inp["k_cache"], inp["v_cache"], inp["rms_w"], meta["eps"],
                                     inp["cos"], inp["sin"], dims=dims)
    # same fp32 formulas -> at most an fp16 rounding flip apart
    assert torch.equal(res, gold["residual"])
    assert max_ulp(k, gold["k_new"]) <= 1 and max_ulp(v, gold["v_new"]) <= 1
    assert max_ulp(out, gold["out"]) <= 1


@pytest.mark.parametrize("name", GPTJ)
def test_oracle_matches_reference_model_gptj(name):
    meta, gold = load_golden(name)
    dims, inp = golden_inputs(meta)
    out, res, k, v = O.decoder_layer(inp["x"], None, inp["weight_qkv"], inp["weight_o"],
                                     inp["k_cache"], inp["v_cache"], inp["rms_w"], 1e-6,
                                     gold["cos"], gold["sin"], dims=dims,
                                     weight_layout="in_out", rope_style="gptj")
    assert res is None
    assert max_ulp(k, gold["k_new"]) <= 1 and max_ulp(v, gold["v_new"]) <= 1
    assert max_ulp(out, gold["out"]) <= 1


@pytest.mark.parametrize("name", ["gqa_gptj_s300", "gqa_gptj_s2100"])
def test_oracle_gqa_matches_reference_model_repeat_kv(name):
    """Grouped-query attention (config 4's geometry, 32 q / 8 kv heads) against fixtures composed from the reference's model.py:
    RMSNorm, apply_rotary_emb and -- the point -- ``repeat_kv`` (chat/llama/model.py:166-175: q head i reads kv head i // 4)."""
    meta, gold = load_golden(name)
    dims, inp = golden_inputs(meta)
    assert O.input_checksum(inp) == meta["input_sha256"], "RNG drift: regenerate goldens"
    out, res, k, v = O.decoder_layer(inp["x"], None, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                     inp["rms_w"], 1e-6, gold["cos"], gold["sin"], dims=dims, weight_layout="out_in", rope_style="gptj")
    assert max_ulp(k, gold["k_new"]) <= 1 and max_ulp(v, gold["v_new"]) <= 1
    assert max_ulp(out, gold["out"]) <= 1


PAGED = ["paged_p1_b6", "paged_p16_b3", "paged_p1_b20"]


def _new_token_slots(inp, meta):
    P, out = meta["page_size"], []
    for b, n_tok in enumerate(meta["lens"]):
        ent = inp["kv_indices"][int(inp["kv_indptr"][b]):int(inp["kv_indptr"][b + 1])].long()
        out.append(int(ent[-1]) if P == 1 else int(ent[n_tok // P]) * P + n_tok % P)
    return out


@pytest.mark.parametrize("name", PAGED)
def test_oracle_paged_batch_matches_reference_eager_row_by_row(name):
    """The paged / batched variant of the oracle against fixtures minted by the reference's own ``reference()``, called once per
    row on the K/V rows its page-table entries name (oracle/gen_golden.py: gen_paged)."""
    meta, gold = load_golden(name)
    inp = O.make_paged_inputs(meta["seed"], meta["page_size"], meta["lens"])
    assert O.input_checksum(inp) == meta["input_sha256"], "RNG drift: regenerate goldens"
    out, res, kc, vc = O.decoder_layer_paged_batch(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"], inp["kv_indptr"],
                                                   inp["kv_indices"], inp["k_cache"], inp["v_cache"], inp["rms_w"], meta["eps"],
                                                   inp["positions"], inp["cos_sin"], page_size=meta["page_size"])
    assert torch.equal(res, gold["residual"])
    slots = _new_token_slots(inp, meta)
    assert max_ulp(kc[slots], gold["k_new"]) <= 1 and max_ulp(vc[slots], gold["v_new"]) <= 1
    assert max_ulp(out, gold["out"]) <= 1
    untouched = torch.ones(kc.shape[0], dtype=torch.bool)
    untouched[slots] = False
    assert torch.equal(kc[untouched], inp["k_cache"][untouched]) and torch.equal(vc[untouched], inp["v_cache"][untouched])


@pytest.mark.parametrize("name", ["gqa_paged_p1_b3", "gqa_paged_p16_b2"])
def test_oracle_gqa_rows_of_a_paged_batch_match_reference_model(name):
    """The grouped-query geometry with several sequences over a paged cache (what k_fused_decode_gb serves): the oracle, row by row on
    the K/V rows the page table names, against fixtures composed from the reference's own model.py helpers (oracle/gen_golden.py:
    gen_gqa_paged) -- RMSNorm, apply_rotary_emb at each row's position, repeat_kv."""
    meta, gold = load_golden(name)
    dims = O.LayerDims(*meta["dims"])
    inp = O.make_paged_inputs(meta["seed"], meta["page_size"], meta["lens"], dims)
    assert O.input_checksum(inp) == meta["input_sha256"], "RNG drift: regenerate goldens"
    P = meta["page_size"]
    for b, n_tok in enumerate(meta["lens"]):
        ent = inp["kv_indices"][int(inp["kv_indptr"][b]):int(inp["kv_indptr"][b + 1])].long()
        slots = ent[:-1] if P == 1 else ent[torch.arange(n_tok) // P] * P + (torch.arange(n_tok) % P)
        out, _, k, v = O.decoder_layer(inp["x"][b:b + 1], None, inp["weight_qkv"], inp["weight_o"], inp["k_cache"][slots], inp["v_cache"][slots],
                                       inp["rms_w"], meta["eps"], gold["cos"][b:b + 1], gold["sin"][b:b + 1], dims=dims, rope_style="gptj")
        assert max_ulp(out.view(-1), gold["out"][b]) <= 1, (b, n_tok)
        assert max_ulp(k.reshape(-1), gold["k_new"][b]) <= 1 and max_ulp(v.reshape(-1), gold["v_new"][b]) <= 1


def test_fp64_and_kernel_rounding_emulation_distances():
    """Distances that justify the tolerances in tests/_util.py."""
    inp = O.make_inputs(42, 128)
    args = (inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"], inp["k_cache"],
            inp["v_cache"], inp["rms_w"], 1e-6, inp["cos"], inp["sin"])
    o32 = O.decoder_layer(*args)
    o64 = O.decoder_layer(*args, compute_dtype=torch.float64)
    oem = O.decoder_layer(*args, emulate_kernel_rounding=True)
    assert max_abs(o32[0], o64[0]) < 2.5e-4
    assert max_abs(oem[0], o64[0]) < 1e-3          # the reference kernel's own rounding noise
    assert max_abs(o32[0], oem[0]) < 1e-3          # build-style vs reference-kernel-style
    # k_new/v_new: the kernel's fp16 partial adds cost ~1 ulp of the LARGEST magnitudes present
    for i in (2, 3):
        assert max_abs(oem[i], o32[i]) <= 2 * ulp16(o32[i].float().abs().max()).item()


def test_layouts_describe_same_map():
    a = O.make_inputs(5, 33, weight_layout="out_in")
    b = O.make_inputs(5, 33, weight_layout="in_out")
    ra = O.decoder_layer(a["x"], None, a["weight_qkv"], a["weight_o"], a["k_cache"], a["v_cache"],
                         a["rms_w"], 1e-6, a["cos"], a["sin"], weight_layout="out_in")
    rb = O.decoder_layer(b["x"], None, b["weight_qkv"], b["weight_o"], b["k_cache"], b["v_cache"],
                         b["rms_w"], 1e-6, b["cos"], b["sin"], weight_layout="in_out")
    assert max_err_in_ulps_of_max(ra[0], rb[0]) <= 1 and max_err_in_ulps_of_max(ra[2], rb[2]) <= 1


def test_gqa_equals_mha_with_repeated_kv():
    dims = O.LayerDims(1024, 8, 2, 128)
    inp = O.make_inputs(3, 50, dims)
    out, _, k, v = O.decoder_layer(inp["x"], None, inp["weight_qkv"], inp["weight_o"],
                                   inp["k_cache"], inp["v_cache"], inp["rms_w"], 1e-6,
                                   inp["cos"], inp["sin"], dims=dims)
    # expand to MHA by repeating the kv projection rows and cache columns
    qd, kd, hd = dims.q_dim, dims.kv_dim, dims.head_dim
    w = inp["weight_qkv"]
    rep = lambda m: m.reshape(2, hd, -1).repeat_interleave(4, 0).reshape(8 * hd, -1)
    w2 = torch.cat([w[:qd], rep(w[qd:qd + kd]), rep(w[qd + kd:])], 0)
    repc = lambda c: c.reshape(-1, 2, hd).repeat_interleave(4, 1).reshape(-1, 8 * hd)
    out2, _, k2, _ = O.decoder_layer(inp["x"], None, w2, inp["weight_o"], repc(inp["k_cache"]),
                                     repc(inp["v_cache"]), inp["rms_w"], 1e-6, inp["cos"],
                                     inp["sin"], dims=O.LayerDims(1024, 8, 8, 128))
    assert max_err_in_ulps_of_max(out, out2) <= 1
    assert torch.equal(k2.reshape(2, 4, hd)[:, 0], k.reshape(2, hd))


def test_tp_shards_sum_to_full():
    dims = O.LayerDims(1024, 8, 8, 128)
    inp = O.make_inputs(9, 40, dims)
    full = O.decoder_layer(inp["x"], inp["residual"], inp["weight_qkv"], inp["weight_o"],
                           inp["k_cache"], inp["v_cache"], inp["rms_w"], 1e-6, inp["cos"],
                           inp["sin"], dims=dims, compute_dtype=torch.float64)
    acc = torch.zeros(1, 1024, dtype=torch.float64)
    ks = []
    for r in range(4):
        wq, wo, kc, vc, d_r = O.shard_for_tp(inp["weight_qkv"], inp["weight_o"], inp["k_cache"],
                                             inp["v_cache"], dims, r, 4)
        o, _, k, _ = O.decoder_layer(inp["x"], inp["residual"], wq, wo, kc, vc, inp["rms_w"], 1e-6,
                                     inp["cos"], inp["sin"], dims=d_r, compute_dtype=torch.float64)
        acc += o.double()
        ks.append(k)
    assert max_abs(acc, full[0]) < 2e-3      # 4 fp16-rounded partials
    assert torch.equal(torch.cat(ks, 1), full[2])


@pytest.mark.parametrize("page_size", [1, 16])
def test_paged_batch_equals_gather_then_single(page_size):
    dims = O.LayerDims(1024, 8, 8, 128)
    g = torch.Generator().manual_seed(11)
    bs, lens = 3, [5, 33, 16]
    inp = O.make_inputs(4, 1, dims)
    x = (torch.randn(bs, 1024, generator=g) * 0.1).half()
    r = (torch.randn(bs, 1024, generator=g) * 0.1).half()
    n_slots = 256
    kc = (torch.randn(n_slots, 1024, generator=g) * 0.1).half()
    vc = (torch.randn(n_slots, 1024, generator=g) * 0.1).half()
    cos_sin = torch.rand(64, 128, generator=g)
    if page_size == 1:
        perm = torch.randperm(n_slots, generator=g)
        counts = [l + 1 for l in lens]
    else:
        perm = torch.randperm(n_slots // page_size, generator=g)
        counts = [(l + 1 + page_size - 1) // page_size for l in lens]
    indptr = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    indices = perm[: indptr[-1]].to(torch.int32)
    positions = torch.tensor(lens, dtype=torch.int64)
    out, res, kc2, vc2 = O.decoder_layer_paged_batch(
        x, r, inp["weight_qkv"], inp["weight_o"], indptr, indices, kc, vc, inp["rms_w"], 1e-6,
        positions, cos_sin, dims=dims, page_size=page_size)
    for b in range(bs):
        ent = indices[indptr[b]:indptr[b + 1]].long()
        t = torch.arange(lens[b] + 1)
        slots = ent[t // page_size] * page_size + t % page_size
        o1, r1, k1, v1 = O.decoder_layer(x[b], r[b], inp["weight_qkv"], inp["weight_o"],
                                         kc[slots[:-1]], vc[slots[:-1]], inp["rms_w"], 1e-6,
                                         cos_sin[lens[b], :64], cos_sin[lens[b], 64:], dims=dims)
        assert torch.equal(out[b:b + 1], o1) and torch.equal(res[b:b + 1], r1)
        assert torch.equal(kc2[slots[-1]], k1.reshape(-1)) and torch.equal(vc2[slots[-1]], v1.reshape(-1))
    untouched = torch.ones(n_slots, dtype=torch.bool)
    for b in range(bs):
        ent = indices[indptr[b]:indptr[b + 1]].long()
        untouched[ent[lens[b] // page_size] * page_size + lens[b] % page_size] = False
    assert torch.equal(kc2[untouched], kc[untouched])


def test_algorithmic_bytes_matches_survey():
    assert abs(O.algorithmic_bytes(O.LLAMA2_7B, 4096) - 201.4e6) < 0.1e6
    assert abs(O.algorithmic_bytes(O.LLAMA2_7B, 128) - 136.4e6) < 0.1e6
    assert abs(O.algorithmic_bytes(O.LLAMA3_8B, 8192) - 117.4e6) < 0.1e6
