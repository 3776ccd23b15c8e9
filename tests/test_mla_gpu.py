"""GPU parity of the DeepSeek MLA decoder-layer op (Python op -> C-ABI -> HIP kernels) against the CPU oracle.

The oracle is "parity unpinned" (the reference has no test / golden for this op, oracle/mla_oracle.py header); the
bar here is the oracle's float64 result of the reference's algorithm.

Tolerance: the op accumulates in fp32 and rounds to fp16 where the matrix cores need fp16 operands (the absorbed
query, the softmax probabilities, the cached latents are fp16 already) and at `out`.  Bound used: 2e-3 x the largest
output magnitude (at least 2e-3 absolute) -- about 2 fp16 ulps of the largest output; the reference's own kernel
rounds at many more points (emulate_kernel_rounding in the oracle differs from exact by up to ~4e-3 on these inputs).
"""
import pytest
import torch

from oracle import mla_oracle as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ORDER = ["input", "weight_q_nope", "weight_q_pe", "weight_uk", "weight_kv_nope", "weight_k_pe", "weight_uv", "weight_o",
         "ckv_cache", "rms_input_weight", "rms_ckv_weight", "cos", "sin"]


@pytest.fixture(scope="module")
def cfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import clusterfusion_amd
    from clusterfusion_amd import _lib
    _lib.load()
    return clusterfusion_amd


@pytest.fixture(params=["pipeline", "fused"], autouse=True)
def path(request, cfa):
    """Both execution paths: three launches, and the single persistent launch ("fused" = required, so a silent
    fall-back cannot pass)."""
    cfa.set_path(request.param)
    yield request.param
    cfa.set_path("auto")
    cfa.check_device_errors()


def _run(cfa, inp, **kw):
    g = [inp[k].to(DEV) for k in ORDER]
    r = cfa.deepseek_decoder_layer(*g, **kw)
    cfa.check_device_errors()        # no in-kernel hand-off gave up
    return r


def _tol(ref):
    return 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("seq_len", [1, 2, 15, 16, 17, 63, 64, 65, 257, 1024, 4096])
@pytest.mark.parametrize("rope_scores", [False, True])
def test_matches_oracle(cfa, seq_len, rope_scores):
    inp = M.make_mla_inputs(100 + seq_len, seq_len, score_gain=3.0)
    ref = M.mla_decoder_layer(inp, rope_scores=rope_scores)
    o, lat = _run(cfa, inp, rope_scores=rope_scores, return_latent=True)
    assert o.shape == (1, 2048) and o.dtype == torch.float16 and lat.shape == (576,)
    err = (o.cpu().double() - ref["o"]).abs().max().item()
    assert err <= _tol(ref["o"]), (err, _tol(ref["o"]))
    lerr = (lat.cpu().double() - ref["latent"]).abs().max().item()
    assert lerr <= 2e-3 * max(1.0, ref["latent"].abs().max().item()), lerr


def test_reference_call_signature_and_default_behaviour(cfa):
    """13 positional tensors -> o [1, 2048] (pybind.cpp:45-59); default = the reference's scores (no rope term)."""
    inp = M.make_mla_inputs(7, 4096, score_gain=3.0)
    o = _run(cfa, inp)
    assert isinstance(o, torch.Tensor) and o.shape == (1, 2048)
    ref = M.mla_decoder_layer(inp)["o"]
    assert (o.cpu().double() - ref).abs().max().item() <= _tol(ref)
    # the rope extension really changes the result on these inputs
    ref_pe = M.mla_decoder_layer(inp, rope_scores=True)["o"]
    assert (ref - ref_pe).abs().max().item() > 10 * _tol(ref)


def test_long_cache_multiple_steps_per_workgroup(cfa):
    """> 16384 entries: every workgroup walks several 64-token steps with the running (m, l) rescale."""
    inp = M.make_mla_inputs(9, 40000, score_gain=4.0)
    ref = M.mla_decoder_layer(inp, rope_scores=True)["o"]
    o = _run(cfa, inp, rope_scores=True)
    assert (o.cpu().double() - ref).abs().max().item() <= _tol(ref)


def test_peaky_and_flat_softmax(cfa):
    for gain in (0.05, 12.0):
        inp = M.make_mla_inputs(21, 777, score_gain=gain)
        ref = M.mla_decoder_layer(inp, rope_scores=True)["o"]
        o = _run(cfa, inp, rope_scores=True)
        assert (o.cpu().double() - ref).abs().max().item() <= _tol(ref), gain


def test_last_row_and_padding_never_read(cfa):
    """NaNs in the new token's slot (and, without rope_scores, in the rope columns) must not reach the output."""
    inp = M.make_mla_inputs(5, 100, score_gain=3.0)
    ref = M.mla_decoder_layer(inp)["o"]
    c = inp["ckv_cache"].clone()
    c[-1] = float("nan")
    c[:, 512:] = float("nan")
    inp2 = dict(inp, ckv_cache=c)
    o = _run(cfa, inp2)
    assert torch.isfinite(o).all()
    assert (o.cpu().double() - ref).abs().max().item() <= _tol(ref)


def test_bit_reproducible_and_workspace_reuse(cfa):
    """Fixed summation orders everywhere (the last-arriver reduction sums in slice order): repeated calls and
    interleaved calls of different lengths give bit-identical results."""
    a = M.make_mla_inputs(1, 3000, score_gain=3.0)
    b = M.make_mla_inputs(2, 50, score_gain=3.0)
    ga, gb = [a[k].to(DEV) for k in ORDER], [b[k].to(DEV) for k in ORDER]
    o1 = cfa.deepseek_decoder_layer(*ga).clone()
    p1 = cfa.deepseek_decoder_layer(*gb).clone()
    for _ in range(20):
        assert torch.equal(cfa.deepseek_decoder_layer(*ga), o1)
        assert torch.equal(cfa.deepseek_decoder_layer(*gb), p1)


def test_linearity_in_output_projection(cfa):
    """Size-independent property at the reference's full size: scaling W_o by 2 scales the output by exactly 2."""
    inp = M.make_mla_inputs(11, 4096, score_gain=3.0)
    o = _run(cfa, inp).float()
    inp2 = dict(inp, weight_o=(inp["weight_o"].float() * 2).half())
    o2 = _run(cfa, inp2).float()
    big = o.abs() >= 2.0 ** -13          # below that fp16 is subnormal and fp16(2 v) != 2 fp16(v) in general
    assert big.float().mean().item() > 0.95
    assert torch.equal(o2[big], o[big] * 2)
    assert (o2 - o * 2).abs().max().item() <= 2.0 ** -23


@pytest.mark.parametrize("seq_len", [64, 1024, 4096])
def test_distance_to_the_reference_kernels_rounding_chain(cfa, seq_len):
    """VERDICT r1 #9 -- the strongest pin available for this op: the reference has no test, golden or eager twin for
    `deepseek_decoder_layer`, so besides the float64 statement of its algorithm the oracle can emulate the CUDA kernel's own
    fp16 rounding points (H100/deepseek/kernel.cuh: normalised activations, 4 split-K partials added in fp16, q_abs, the
    attention partials, fp16 accumulation over heads).  Three-way bound on the same inputs:
        |GPU - exact|   <= 2e-3 x max|out|      (what test_matches_oracle holds)
        |GPU - kernel-rounding emulation| <= |emulation - exact| + 2e-3 x max|out|
    i.e. the GPU result is at least as close to the reference kernel's result as the exact value is, up to our own two
    ulps -- it sits inside the ball the reference kernel's rounding noise draws around the exact result."""
    inp = M.make_mla_inputs(300 + seq_len, seq_len, score_gain=3.0)
    exact = M.mla_decoder_layer(inp)["o"]
    emu = M.mla_decoder_layer(inp, emulate_kernel_rounding=True)["o"]
    o = _run(cfa, inp).cpu().double()
    scale = max(1.0, exact.abs().max().item())
    d_exact = (o - exact).abs().max().item()
    d_emu = (o - emu).abs().max().item()
    d_ref_noise = (emu - exact).abs().max().item()
    assert d_exact <= 2e-3 * scale, (d_exact, scale)
    assert d_emu <= d_ref_noise + 2e-3 * scale, (d_emu, d_ref_noise, scale)
    assert d_ref_noise <= 2e-2 * scale          # the emulation itself stays a rounding-level perturbation


@pytest.mark.parametrize("seq_len", [4096, 20000])
def test_large_magnitude_latents_do_not_overflow_fp16_partials(cfa, seq_len):
    """the persistent kernel publishes its attention partials as fp16 pairs.  They go out NORMALISED per unit
    (O_s / l_s, bounded by max |v|): latents of magnitude ~600 with near-uniform attention, whose un-normalised sums over a
    unit's 128+ tokens exceed 65504, must still come out finite and right."""
    inp = M.make_mla_inputs(51, seq_len, score_gain=0.0)
    g = torch.Generator().manual_seed(5)
    big = 600.0 + 8.0 * torch.randn(seq_len, 512, generator=g)
    inp["ckv_cache"] = torch.cat([big, inp["ckv_cache"][:, 512:].float()], dim=1).half()
    ref = M.mla_decoder_layer(inp)["o"]
    o = _run(cfa, inp)
    assert torch.isfinite(o).all()
    err = (o.cpu().double() - ref).abs().max().item()
    assert err <= _tol(ref), (err, _tol(ref))
