"""The compiled host binding (clusterfusion_amd/_cf_fast, csrc/cf_torch_binding.cpp): the reference binds its entries as
direct C++ functions (include/pybind.cpp:108-112); ours does the same above the C-ABI.

CPU half: the module is built, imports, and DECLINES (NotImplemented) what it must not launch -- the Python entries then
raise with their own messages.  GPU half: the three entries are served by the binding (its counters say so), bit-identical to
the ctypes path, across weight updates, stream switches, capture, and a reported exchange failure."""
import os

import pytest
import torch

from oracle import cf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _built():
    """Both shared objects are build artefacts (git-ignored): make sure they exist before anything imports them."""
    from clusterfusion_amd import build as cfbuild
    cfbuild.build()


def test_binding_is_built_and_declines_cpu_tensors():
    import clusterfusion_amd as cfa
    from clusterfusion_amd import _cf_fast as f
    assert os.path.exists(os.path.join(ROOT, "clusterfusion_amd", "_cf_fast.so"))
    assert cfa.host_binding() == "compiled"
    x = torch.zeros(4096, dtype=torch.float16)
    before = f.stats()
    assert f.llama_decoder_layer(x, x, x, x, x, x, x, x) is NotImplemented
    assert f.llama_decoder_layer_sglang(x, x, x, x, x, x, x, 1e-6, x, x) is NotImplemented
    assert f.llama_decoder_layer_batch_decode_sglang(x, x, x, x, x, x, x, x, x, x, 0, x, 1e-6, x, x) is NotImplemented
    after = f.stats()
    assert after[0] == before[0] and after[1] == before[1] + 3


def test_python_entries_keep_their_error_messages_with_the_binding_on():
    import clusterfusion
    x = torch.zeros(4096, dtype=torch.float16)
    with pytest.raises(ValueError, match="must live on the GPU"):
        clusterfusion.llama_decoder_layer_sglang(x, x, x, x, x, x, x, 1e-6, x, x)
    with pytest.raises(TypeError, match="expected a torch.Tensor"):
        clusterfusion.llama_decoder_layer([1, 2], x, x, x, x, x, x, x)
    with pytest.raises(TypeError, match="expected dtype"):
        clusterfusion.llama_decoder_layer_sglang(x.float(), x, x, x, x, x, x, 1e-6, x, x)


def test_host_binding_switch():
    import clusterfusion_amd as cfa
    cfa.set_host_binding("ctypes")
    try:
        assert cfa.host_binding() == "ctypes"
        assert cfa.host_binding_stats()["taken"] == 0
    finally:
        cfa.set_host_binding("compiled")
    assert cfa.host_binding() == "compiled"
    with pytest.raises(ValueError):
        cfa.set_host_binding("jit")


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture()
def cfa():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import clusterfusion_amd
    clusterfusion_amd.set_host_binding("compiled")
    yield clusterfusion_amd
    clusterfusion_amd.set_host_binding("compiled")
    clusterfusion_amd.check_device_errors()


def _gpu(inp):
    return {k: v.to(DEV) for k, v in inp.items()}


def _taken(cfa):
    return cfa.host_binding_stats()["taken"]


@pytest.mark.gpu
def test_sglang_entry_is_served_by_the_binding_bit_identical_to_ctypes(cfa):
    import clusterfusion
    inp = _gpu(O.make_inputs(301, 777))
    cpu = O.make_inputs(301, 777)

    def call():
        res = inp["residual"].clone()
        return clusterfusion.llama_decoder_layer_sglang(inp["x"], res, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                                        inp["rms_w"], 1e-6, inp["cos"], inp["sin"]), res
    call()                                   # (first call on this stream: the Python entry sets the workspace up)
    t0 = _taken(cfa)
    (o, r, k, v), res = call()
    assert _taken(cfa) == t0 + 1, "the compiled binding did not take the call"
    assert r is res and o.shape == (1, 4096) and k.shape == (1, 32, 128) and v.shape == (1, 32, 128)
    cfa.set_host_binding("ctypes")
    (o2, r2, k2, v2), _ = call()
    assert cfa.host_binding_stats()["taken"] == 0
    torch.cuda.synchronize()
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(k, k2) and torch.equal(v, v2)
    ro, rr, rk, rv = O.decoder_layer(cpu["x"], cpu["residual"], cpu["weight_qkv"], cpu["weight_o"], cpu["k_cache"], cpu["v_cache"],
                                     cpu["rms_w"], 1e-6, cpu["cos"], cpu["sin"])
    assert (o.cpu().float() - ro.float()).abs().max().item() <= 1e-3
    assert torch.equal(r.cpu(), rr)


@pytest.mark.gpu
def test_plain_entry_binding_follows_weight_updates_and_capture(cfa):
    import clusterfusion
    inp = _gpu(O.make_inputs(302, 300, weight_layout="in_out"))
    ang = torch.rand(64, generator=torch.Generator().manual_seed(5)) * 6.28
    cos = ang.cos().repeat_interleave(2).view(1, 128).contiguous().to(DEV)
    sin = ang.sin().repeat_interleave(2).view(1, 128).contiguous().to(DEV)
    wq, wo = inp["weight_qkv"], inp["weight_o"]

    def call():
        return clusterfusion.llama_decoder_layer(inp["x"].view(1, 1, 4096), wq, wo, inp["k_cache"], inp["v_cache"], inp["rms_w"], cos, sin)
    cfa.set_weight_relayout(True)
    try:
        call()                               # makes the [out,in] copy (Python entry)
        t0 = _taken(cfa)
        o1, k1, v1 = call()
        assert _taken(cfa) == t0 + 1 and cfa.last_variant() == "k_fused_decode_mha<IO=false>"
        cfa.set_host_binding("ctypes")
        o2, k2, v2 = call()
        cfa.set_host_binding("compiled")
        torch.cuda.synchronize()
        assert torch.equal(o1, o2) and torch.equal(k1, k2) and torch.equal(v1, v2)
        # an in-place update moves the version counter: the binding must decline, the Python entry re-lays out, results change
        wo.mul_(0.5)
        t0 = _taken(cfa)
        o3, _, _ = call()
        assert _taken(cfa) == t0, "the binding served a call from a stale weight copy"
        o4, _, _ = call()                    # ... and serves the refreshed copy again
        assert _taken(cfa) == t0 + 1
        torch.cuda.synchronize()
        assert torch.equal(o3, o4) and not torch.equal(o3, o1)
        assert (o3.float() - 0.5 * o1.float()).abs().max().item() <= 2e-3
        # capture: the binding hands the call to the Python entry (which marks the copy as seen by a graph); replays are right
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            call()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            t0 = _taken(cfa)
            with torch.cuda.graph(g, stream=s):
                og, _, _ = call()
            assert _taken(cfa) == t0
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(og, o3)
        assert cfa.weight_relayout_stats()["pinned_by_capture"] >= 1
        # re-layout off: nothing registered, the native [in,out] kernel runs through the Python entry
        cfa.set_weight_relayout(False)
        assert cfa.host_binding_stats()["weight_copies"] == 0
        t0 = _taken(cfa)
        o5, _, _ = call()
        assert _taken(cfa) == t0 and cfa.last_variant() == "k_fused_decode_mha<IO=true>"
        torch.cuda.synchronize()
        assert (o5.float() - o3.float()).abs().max().item() <= 1e-3
    finally:
        cfa.set_weight_relayout(True)
        cfa.release_weight_relayout()


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [1, 3, 8])
def test_batch_entry_is_served_by_the_binding_bit_identical_to_ctypes(cfa, bs):
    import clusterfusion
    g = torch.Generator().manual_seed(400 + bs)
    S = [37, 260, 5, 129, 64, 1, 300, 17][:bs]
    n_slots = sum(s + 1 for s in S) + 9
    rn = lambda *sh: (torch.randn(*sh, generator=g) * 0.1).half().to(DEV)      # noqa: E731
    kc, vc = rn(n_slots, 4096), rn(n_slots, 4096)
    wq, wo, rms = rn(12288, 4096), rn(4096, 4096), rn(4096)
    perm = torch.randperm(n_slots, generator=g).to(torch.int32)
    indptr = torch.zeros(bs + 1, dtype=torch.int32)
    for b, s in enumerate(S):
        indptr[b + 1] = indptr[b] + s + 1
    indices = perm[: int(indptr[-1])].contiguous().to(DEV)
    indptr = indptr.to(DEV)
    positions = torch.tensor(S, dtype=torch.int64, device=DEV)
    cos_sin = (torch.rand(512, 128, generator=g) * 2 - 1).float().to(DEV)
    x, r = rn(bs, 4096), rn(bs, 4096)
    kptrs = torch.tensor([0, kc.data_ptr()], dtype=torch.uint64, device=DEV)
    vptrs = torch.tensor([0, vc.data_ptr()], dtype=torch.uint64, device=DEV)

    def call():
        k0, v0 = kc.clone(), vc.clone()
        kp = torch.tensor([0, k0.data_ptr()], dtype=torch.uint64, device=DEV)
        vp = torch.tensor([0, v0.data_ptr()], dtype=torch.uint64, device=DEV)
        o, ro = torch.empty_like(x), torch.empty_like(x)
        assert clusterfusion.llama_decoder_layer_batch_decode_sglang(o, ro, x, r, wq, wo, indptr, indices, kp, vp, 1, rms, 1e-6, positions,
                                                                     cos_sin) is None
        torch.cuda.synchronize()
        return o, ro, k0, v0
    del kptrs, vptrs
    call()
    t0 = _taken(cfa)
    a = call()
    assert _taken(cfa) == t0 + 1
    cfa.set_host_binding("ctypes")
    b = call()
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb)
    assert not torch.equal(a[2], kc)         # (the new token's slot was written)


@pytest.mark.gpu
def test_binding_on_a_second_stream_sets_up_through_python_first(cfa):
    import clusterfusion
    inp = _gpu(O.make_inputs(303, 129))

    def call():
        res = inp["residual"].clone()
        return clusterfusion.llama_decoder_layer_sglang(inp["x"], res, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                                        inp["rms_w"], 1e-6, inp["cos"], inp["sin"])[0]
    ref = call()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t0, w0 = _taken(cfa), cfa.host_binding_stats()["workspaces"]
        o1 = call()                          # no workspace for this stream yet: declined, Python sets it up and registers it
        assert _taken(cfa) == t0 and cfa.host_binding_stats()["workspaces"] == w0 + 1
        o2 = call()
        assert _taken(cfa) == t0 + 1
        s.synchronize()
    assert torch.equal(o1, ref) and torch.equal(o2, ref)


@pytest.mark.gpu
def test_binding_raises_the_sticky_exchange_failure(cfa):
    """A persistent launch that lost co-residency is reported by the NEXT call (DESIGN: sticky word) -- also when that call
    comes through the compiled binding: it must raise CFError, not hand the call to Python (which would launch and hide it)."""
    import clusterfusion
    from clusterfusion_amd import _lib
    lib = _lib.load()
    inp = _gpu(O.make_inputs(304, 1500))

    def call():
        res = inp["residual"].clone()
        return clusterfusion.llama_decoder_layer_sglang(inp["x"], res, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                                        inp["rms_w"], 1e-6, inp["cos"], inp["sin"])[0]
    ref = call()
    torch.cuda.synchronize()
    cfa.check_device_errors()
    side = torch.cuda.Stream()
    lib.cf_debug_occupy(side.cuda_stream, 96, 100 * 1024, 3_000_000)
    t0 = _taken(cfa)
    o_b = call()
    assert _taken(cfa) == t0 + 1
    torch.cuda.synchronize()
    good = torch.equal(o_b, ref)
    raised = False
    try:
        call()
    except _lib.CFError as e:
        raised = True
        assert "co-resident" in str(e)
    assert good or raised, "a failed persistent launch went unreported through the compiled binding"
    torch.cuda.synchronize()
    if raised:
        with pytest.raises(_lib.CFError):
            cfa.check_device_errors()
    cfa.check_device_errors()
    o_d = call()
    torch.cuda.synchronize()
    assert torch.equal(o_d, ref)


@pytest.mark.gpu
def test_plain_entry_with_inference_mode_weights(cfa):
    """Weights made under torch.inference_mode() (how serving code loads them) carry no version counter: the re-layout cache and the
    compiled binding must treat them as version 0 instead of raising (round 6: both used to)."""
    import clusterfusion
    inp = O.make_inputs(305, 200, weight_layout="in_out")
    ang = torch.rand(64, generator=torch.Generator().manual_seed(6)) * 6.28
    with torch.inference_mode():
        g = {k: v.to(DEV) for k, v in inp.items()}
        cos = ang.cos().repeat_interleave(2).view(1, 128).contiguous().to(DEV)
        sin = ang.sin().repeat_interleave(2).view(1, 128).contiguous().to(DEV)
        assert g["weight_qkv"].is_inference()

        def call():
            return clusterfusion.llama_decoder_layer(g["x"].view(1, 1, 4096), g["weight_qkv"], g["weight_o"], g["k_cache"], g["v_cache"], g["rms_w"], cos, sin)
        cfa.set_weight_relayout(True)
        try:
            o1, k1, v1 = call()
            t0 = _taken(cfa)
            o2, k2, v2 = call()
            assert _taken(cfa) == t0 + 1
            torch.cuda.synchronize()
            assert torch.equal(o1, o2) and torch.equal(k1, k2)
        finally:
            cfa.release_weight_relayout()
    ro, _, rk, rv = O.decoder_layer(inp["x"], None, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"], inp["rms_w"], 1e-6,
                                    ang.cos().repeat_interleave(2).view(1, 128), ang.sin().repeat_interleave(2).view(1, 128),
                                    weight_layout="in_out", rope_style="gptj")
    assert (o1.cpu().float() - ro.float()).abs().max().item() <= 1e-3


@pytest.mark.gpu
def test_sglang_style_entries_take_the_grouped_query_geometry(cfa):
    """Extension (BASELINE config 4 through the reference's own entry names): weight_qkv [6144, 4096] = 32 q / 8 kv heads.  The
    single-sequence entry runs k_fused_decode_g<8, 4>, the batched one k_fused_decode_gb; both against the oracle (repeat_kv,
    chat/llama/model.py:166-175), both served by the compiled binding from the second call on, bit-identical to the ctypes path."""
    import clusterfusion
    dims = O.LayerDims(4096, 32, 8, 128)
    cpu = O.make_inputs(306, 900, dims)
    inp = _gpu(cpu)

    def call():
        res = inp["residual"].clone()
        return clusterfusion.llama_decoder_layer_sglang(inp["x"], res, inp["weight_qkv"], inp["weight_o"], inp["k_cache"], inp["v_cache"],
                                                        inp["rms_w"], 1e-6, inp["cos"], inp["sin"]), res
    call()
    assert cfa.last_variant() == "k_fused_decode_g<8, 4>"
    t0 = _taken(cfa)
    (o, r, k, v), res = call()
    assert _taken(cfa) == t0 + 1 and r is res and k.shape == (1, 8, 128) and v.shape == (1, 8, 128)
    cfa.set_host_binding("ctypes")
    (o2, r2, k2, v2), _ = call()
    cfa.set_host_binding("compiled")
    torch.cuda.synchronize()
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(k, k2) and torch.equal(v, v2)
    ro, rr, rk, rv = O.decoder_layer(cpu["x"], cpu["residual"], cpu["weight_qkv"], cpu["weight_o"], cpu["k_cache"], cpu["v_cache"],
                                     cpu["rms_w"], 1e-6, cpu["cos"], cpu["sin"], dims=dims)
    assert (o.cpu().float() - ro.float()).abs().max().item() <= 1e-3 and torch.equal(r.cpu(), rr)
    # ---- the batched entry: 3 rows, page size 1 (the reference's semantics), caches [num_slots, 1024] behind pointer tables
    pin = O.make_paged_inputs(307, 1, [400, 0, 2300], dims)
    g = {k_: v_.to(DEV) for k_, v_ in pin.items()}
    eo, er, ekc, evc = O.decoder_layer_paged_batch(pin["x"], pin["residual"], pin["weight_qkv"], pin["weight_o"], pin["kv_indptr"], pin["kv_indices"],
                                                   pin["k_cache"], pin["v_cache"], pin["rms_w"], 1e-6, pin["positions"], pin["cos_sin"], dims=dims)

    def bcall():
        kc, vc = g["k_cache"].clone(), g["v_cache"].clone()
        kp = torch.tensor([0, kc.data_ptr()], dtype=torch.uint64, device=DEV)
        vp = torch.tensor([0, vc.data_ptr()], dtype=torch.uint64, device=DEV)
        out, rout = torch.empty_like(g["x"]), torch.empty_like(g["x"])
        assert clusterfusion.llama_decoder_layer_batch_decode_sglang(out, rout, g["x"], g["residual"], g["weight_qkv"], g["weight_o"], g["kv_indptr"],
                                                                     g["kv_indices"], kp, vp, 1, g["rms_w"], 1e-6, g["positions"], g["cos_sin"]) is None
        torch.cuda.synchronize()
        return out, rout, kc, vc
    bcall()
    assert cfa.last_variant() == "k_fused_decode_gb<4>"
    t0 = _taken(cfa)
    a = bcall()
    assert _taken(cfa) == t0 + 1
    cfa.set_host_binding("ctypes")
    b = bcall()
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb)
    for row in range(3):
        tol = max(1e-3, float(2.0 ** (torch.floor(torch.log2(eo[row].float().abs().max())) - 10)))
        assert (a[0][row].cpu().float() - eo[row].float()).abs().max().item() <= tol
    assert torch.equal(a[1].cpu(), er)
    assert ((a[2].cpu().float() - ekc.float()).abs().max().item() <= 2e-3) and ((a[3].cpu().float() - evc.float()).abs().max().item() <= 2e-3)
