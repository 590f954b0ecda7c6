"""Stage-level C entry points (ivlm_llama_prefill / ivlm_llama_decode_step) through ctypes against the
Python-sequenced path of interactvlm_amd/llava.py: same kernels, same order -> bit-identical hidden states and KV cache."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden,heads,inter,fuse", [(1024, 8, 1376, False), (1024, 8, 1376, True), (256, 2, 512, False)])
def test_llama_prefill_and_decode_step_equal_python_path(hip_lib, cuda, hidden, heads, inter, fuse):
    import torch

    from interactvlm_amd import llava, stages
    from interactvlm_amd import weights as Wt

    lc = Wt.LlamaCfg(hidden=hidden, layers=3, heads=heads, inter=inter, vocab=1000)
    w = {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(Wt.llama_spec(lc)).items()}
    g = torch.Generator().manual_seed(5)
    T0, n_new = 330 if hidden == 1024 else 41, 6
    emb = (torch.randn(T0 + n_new, hidden, generator=g) * 0.5).to(torch.bfloat16).float().to(cuda)
    # Python-sequenced reference: prefill, then decode steps through the captured graph (fused attention + o_proj where it applies)
    a = llava.Llama(w, lc, cuda, max_len=512)
    a.decode_packed = False  # (the C sequencers of this test stream the bf16 weights: compare like with like, bit for bit)
    a.fuse_attn_oproj = fuse  # (attention + o_proj in one launch: opt-in on both sides)
    ha = [a.forward(emb[:T0], 0)]
    dg = a.decode_graph()
    assert (dg.get("fused") is not None) == fuse
    dg["pos"].fill_(T0)
    if dg.get("fused") is not None:
        for k in ("step", "counters", "status"):
            dg["fused"][k].zero_()
    pos = torch.tensor([T0], dtype=torch.int32, device=cuda)
    for t in range(n_new):
        ha.append(a._decode_step(emb[T0 + t: T0 + t + 1], pos if dg.get("fused") is None else dg["pos"]))
        if dg.get("fused") is not None:
            dg["pos"].add_(1)
            dg["fused"]["step"].add_(1)
        else:
            pos.add_(1)
    # C sequencers on a second instance (its own KV cache)
    b = llava.Llama(w, lc, cuda, max_len=512)
    b.fuse_attn_oproj = fuse
    st = stages.LlamaStages(b)
    hb = [st.prefill(emb[:T0], 0)]
    st.start_generation()
    posb = torch.tensor([T0], dtype=torch.int32, device=cuda)
    for t in range(n_new):
        hb.append(st.decode_step(emb[T0 + t: T0 + t + 1].contiguous(), posb, advance=True))
    assert int(posb[0]) == T0 + n_new
    assert torch.equal(hb[0], ha[0]), float((hb[0] - ha[0]).abs().max())
    for t in range(n_new):
        assert torch.equal(hb[1 + t], ha[1 + t]), (t, float((hb[1 + t] - ha[1 + t]).abs().max()))
    n = T0 + n_new
    assert torch.equal(b.kcache[:, :n], a.kcache[:, :n]) and torch.equal(b.vcache[:, :n], a.vcache[:, :n])
    if not fuse:
        # the DEFAULT precision of the host model: ivlm_llama_prefill_f16 / ivlm_llama_decode_step_f16kv (fp16 MFMA operands, fp16
        # KV cache) == the Python-sequenced "f16" path bit for bit
        a2 = llava.Llama(w, lc, cuda, max_len=512)
        a2.decode_packed = False
        a2.set_precision("f16")
        h2 = [a2.forward(emb[:T0], 0)]
        for t in range(n_new):
            h2.append(a2.forward(emb[T0 + t: T0 + t + 1], T0 + t))
        b2 = llava.Llama(w, lc, cuda, max_len=512)
        b2.set_precision("f16")  # (only so that its cache view is the fp16 one below)
        st2 = stages.LlamaStages(b2)
        g2 = [st2.prefill_f16(emb[:T0], 0)]
        st2.start_generation()
        pos2 = torch.tensor([T0], dtype=torch.int32, device=cuda)
        for t in range(n_new):
            g2.append(st2.decode_step_f16kv(emb[T0 + t: T0 + t + 1].contiguous(), pos2, advance=True))
        for t in range(n_new + 1):
            assert torch.equal(g2[t], h2[t]), (t, float((g2[t] - h2[t]).abs().max()))
        ka, kb = a2._caches()[0], b2._caches()[0]
        assert ka.dtype == torch.float16 and torch.equal(kb[:, :n], ka[:, :n]) and torch.equal(b2._caches()[1][:, :n], a2._caches()[1][:, :n])
        assert not torch.equal(h2[0], ha[0])  # (and it is a different rounding than the bf16 path)
    # errors: a workspace that is too small is reported, not overrun
    import ctypes as C
    from interactvlm_amd import _lib
    lib = _lib.load()
    rc = lib.ivlm_llama_decode_step(C.byref(st.cfg), st.layers, b.norm.data_ptr(), b.kcache.data_ptr(), b.vcache.data_ptr(),
                                    b.rope[0].data_ptr(), b.rope[1].data_ptr(), emb.data_ptr(), posb.data_ptr(), 0,
                                    hb[0].data_ptr(), st._dws.data_ptr(), 1024, None)
    assert rc == -2


def test_clip_encode_stage_equals_python_path(hip_lib, cuda):
    import torch

    from interactvlm_amd import llava, stages
    from interactvlm_amd import weights as Wt

    cc = Wt.ClipCfg(hidden=256, layers=4, heads=4, inter=512)
    w = Wt.synth_weights(Wt.clip_spec(cc))
    tower = llava.ClipTower(w, cc, cuda)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 3, 224, 224, generator=g).to(torch.bfloat16).to(cuda)
    ref = tower._forward(x)
    got = stages.ClipStages(tower)(x)
    assert got.shape == ref.shape and torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    # the default precision of the host model: ivlm_clip_encode_f16 == ClipTower in "f16" precision (split feature rows)
    tower.precision = "f16"
    ref16 = tower._forward(x)
    got16 = stages.ClipStages(tower)(x, precision="f16")
    assert got16.shape == ref16.shape == (3, 256, 512) and torch.equal(got16, ref16)
    assert not torch.equal(ref16[..., :256], ref)


@pytest.mark.parametrize("V", [1, 4])
def test_sam_encode_stage_equals_python_path(hip_lib, cuda, V):
    """ivlm_sam_encode (device-built window maps, scatter / gather row maps in the q|k|v and proj GEMMs, rel-pos through the
    batched GEMM for global blocks and the dot kernel for windows, neck) == SamImageEncoder._forward bit for bit, at the real
    ViT-H layer dimensions (2 windowed + 2 global blocks)."""
    import torch

    from interactvlm_amd import sam, stages
    from interactvlm_amd import weights as Wt

    c = Wt.SamEncCfg(depth=4, global_attn_indexes=(1, 3))
    w = Wt.synth_weights(Wt.sam_encoder_spec(c))
    enc = sam.SamImageEncoder(w, c, cuda)
    g = torch.Generator().manual_seed(V)
    x = torch.randn(V, 3, 1024, 1024, generator=g).to(torch.bfloat16).to(cuda)
    ref = enc._forward(x)
    got = stages.SamEncodeStages(enc)(x)
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # "parity" precision: ivlm_sam_encode_parity == SamImageEncoder._forward_parity bit for bit (and both differ from the default)
    enc.precision = "parity"
    ref_p = enc._forward(x)
    got_p = stages.SamEncodeStages(enc)(x, precision="parity")
    assert torch.equal(got_p, ref_p), float((got_p - ref_p).abs().max())
    assert not torch.equal(ref_p, ref)
    # "parity-encoder": the MLP GEMMs on fp16 operands - ivlm_sam_encode_parity_f16mlp == the Python path with PARITY_SITES_FAST
    enc.parity_sites = enc.PARITY_SITES_FAST
    ref_f = enc._forward(x)
    got_f = stages.SamEncodeStages(enc)(x, precision="parity-encoder")
    assert torch.equal(got_f, ref_f), float((got_f - ref_f).abs().max())
    assert not torch.equal(ref_f, ref_p)
    assert float((ref_f - ref_p).abs().max()) < 0.25 * float((ref - ref_p).abs().max())  # (much closer to all-split than default is)
    # the DEFAULT precision of the host model (fp16 operands, exact q path): ivlm_sam_encode_f16 == the Python path with SITES_F16Q
    enc.parity_sites = enc.SITES_F16Q
    ref_q = enc._forward(x)
    got_q = stages.SamEncodeStages(enc)(x, precision="f16")
    assert torch.equal(got_q, ref_q), float((got_q - ref_q).abs().max())
    assert float((ref_q - ref_p).abs().max()) < 0.25 * float((ref - ref_p).abs().max())


@pytest.mark.parametrize("V", [4, 1])
def test_sam_decode_stage_equals_python_path(hip_lib, cuda, V):
    """ivlm_sam_decode (the whole fp32-activation prompt-encoder / two-way-transformer / upscaler / hypernetwork chain as one C
    call) == SamMaskDecoder._forward bit for bit."""
    import torch

    from interactvlm_amd import sam, stages, synth
    from interactvlm_amd import weights as Wt

    w = Wt.synth_weights({**Wt.prompt_encoder_spec(), **Wt.mask_decoder_spec()})
    dec = sam.SamMaskDecoder(w, cuda)
    emb = torch.from_numpy(synth.synth_normal(f"samdec/image_emb/{V}", (V, 256, 64, 64), 1.0, 0))
    emb = emb.permute(0, 2, 3, 1).reshape(V, 4096, 256).contiguous().to(cuda)
    text = torch.from_numpy(synth.synth_normal(f"samdec/text/{V}", (1, V, 256), 1.0, 0)).to(cuda)
    low_r, iou_r = dec._forward(emb, text)
    low, iou = stages.SamDecodeStages(dec)(emb, text)
    assert low.shape == low_r.shape and iou.shape == iou_r.shape
    assert torch.equal(low, low_r), float((low - low_r).abs().max())
    assert torch.equal(iou, iou_r), float((iou - iou_r).abs().max())


@pytest.mark.parametrize("precision", ["f16", "default"])
def test_llama_decode_step_bf12_equals_python_packed_path(hip_lib, cuda, precision):
    """ivlm_llama_decode_step_bf12 (the four linears of a layer on losslessly packed weights, dots on the matrix cores) == the host
    model's default decode path (`decode_packed`) bit for bit: hidden states and the appended KV rows, fp16 and bf16 cache; a
    configuration whose matrices do not take the fragment layout is refused."""
    import ctypes as C

    import torch

    from interactvlm_amd import llava, stages
    from interactvlm_amd import weights as Wt

    lc = Wt.LlamaCfg(hidden=1024, heads=8, layers=3, inter=2752, vocab=1000)  # (2752 = 43 x 64: uneven step pairs over the waves)
    w = {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(Wt.llama_spec(lc)).items()}
    g = torch.Generator().manual_seed(9)
    T0, n_new = 70, 5
    emb = (torch.randn(T0 + n_new, 1024, generator=g) * 0.5).to(torch.bfloat16).float().to(cuda)
    a = llava.Llama(w, lc, cuda, max_len=128)
    a.set_precision(precision)
    assert a.decode_packed
    ha = [a.forward(emb[:T0], 0)] + [a.forward(emb[T0 + t: T0 + t + 1], T0 + t) for t in range(n_new)]
    assert all(a.layers[0][n + "_p"].frag for n in ("qkv", "o", "gu", "down"))
    b = llava.Llama(w, lc, cuda, max_len=128)
    b.set_precision(precision)
    st = stages.LlamaStages(b)
    hb = [st.prefill_f16(emb[:T0], 0) if precision == "f16" else st.prefill(emb[:T0], 0)]
    st.start_generation()
    pos = torch.tensor([T0], dtype=torch.int32, device=cuda)
    for t in range(n_new):
        hb.append(st.decode_step_bf12(emb[T0 + t: T0 + t + 1].contiguous(), pos, advance=True))
    assert int(pos[0]) == T0 + n_new
    for t in range(n_new + 1):
        assert torch.equal(hb[t], ha[t]), (t, float((hb[t] - ha[t]).abs().max()))
    n = T0 + n_new
    (ka, va), (kb, vb) = a._caches(), b._caches()
    assert torch.equal(ka[:, :n], kb[:, :n]) and torch.equal(va[:, :n], vb[:, :n])
    # the bf16-weight step gives the same numbers up to the fp32 summation order (lossless packing, exact products on both)
    c2 = llava.Llama(w, lc, cuda, max_len=128)
    c2.set_precision(precision)
    c2.decode_packed = False
    hc = [c2.forward(emb[:T0], 0)] + [c2.forward(emb[T0 + t: T0 + t + 1], T0 + t) for t in range(n_new)]
    assert float((torch.cat(hc) - torch.cat(ha)).abs().max()) < 2e-5 * float(torch.cat(ha).abs().max())
    # inter % 64 != 0: refused, not mis-read
    bad = stages.LlamaCfg(lc.layers, 1024, 8, 1376, 128, lc.eps, lc.theta, 0)
    rc = hip_lib.ivlm_llama_decode_step_bf12(C.byref(bad), st._layers_bf12(), b.norm.data_ptr(), b.kcache.data_ptr(), b.vcache.data_ptr(),
                                             1, b.rope[0].data_ptr(), b.rope[1].data_ptr(), emb.data_ptr(), pos.data_ptr(), 0,
                                             hb[0].data_ptr(), st._dws.data_ptr(), st._dws.numel(), 0)
    assert rc != 0


def test_stage_splitk_fused_switch_is_opt_in_and_bit_identical(hip_lib, cuda):
    """the C sequencers take the two-launch split-K form unless ivlm_stages_splitk_fused(1) (or IVLM_SPLITK_FUSED=1)
    asks for the fused reduction; both forms give the same bits (the fused one sums a tile's slices in slice order too)."""
    import os

    import torch

    from interactvlm_amd import llava, stages
    from interactvlm_amd import weights as Wt

    if os.environ.get("IVLM_SPLITK_FUSED") != "1":
        assert hip_lib.ivlm_stages_splitk_fused(-1) == 0  # a query: off by default
    lc = Wt.LlamaCfg(hidden=1024, layers=2, heads=8, inter=1376, vocab=1000)
    w = {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(Wt.llama_spec(lc)).items()}
    g = torch.Generator().manual_seed(9)
    emb = (torch.randn(330, 1024, generator=g) * 0.5).to(torch.bfloat16).float().to(cuda)
    outs = []
    prev = hip_lib.ivlm_stages_splitk_fused(-1)
    try:
        for on in (0, 1, 0):
            hip_lib.ivlm_stages_splitk_fused(on)
            assert hip_lib.ivlm_stages_splitk_fused(-1) == on
            b = llava.Llama(w, lc, cuda, max_len=512)
            outs.append((stages.LlamaStages(b).prefill(emb, 0).clone(), b.kcache[:, :330].clone()))
    finally:
        hip_lib.ivlm_stages_splitk_fused(prev)
    for h, k in outs[1:]:
        assert torch.equal(h, outs[0][0]) and torch.equal(k, outs[0][1])
