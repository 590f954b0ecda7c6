"""GPU parity of the fused attention kernel against plain PyTorch fp32 softmax(QK^T)V on the same
bf16-rounded inputs (the HIP kernel keeps softmax in fp32 and rounds P to bf16 for the PV MFMA)."""
import math

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["4-wave", "ping-pong"])
def attn_block_shape(request, hip_lib):
    """Every case runs through both block shapes: 4 waves / 128 queries, and the 8-wave ping-pong block (256 queries, the
    two wave groups half a tile apart: edge tiles, causal diagonals and the drain of the trailing group differ)."""
    from interactvlm_amd import _lib

    lib = _lib.load()
    lib.ivlm_attention_pingpong(0 if request.param == "4-wave" else 1)
    yield request.param
    lib.ivlm_attention_pingpong(-1)


def _ref(q, k, v, scale, causal=False, q_pos0=0, bias=None):
    import torch

    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) * scale
    if bias is not None:
        s = s + bias
    if causal:
        Sq, Sk = q.shape[2], k.shape[2]
        qi = torch.arange(Sq)[:, None] + q_pos0
        kj = torch.arange(Sk)[None, :]
        s = s.masked_fill(kj > qi, float("-inf"))
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, dim=-1), v.float())


CASES = [  # B, H, Sq, Sk, D, causal, q_pos0
    (1, 8, 9, 9, 32, False, 0), (4, 8, 9, 4096, 16, False, 0), (4, 8, 4096, 9, 16, False, 0),
    (3, 16, 196, 196, 80, False, 0), (1, 16, 257, 257, 64, False, 0), (1, 32, 330, 330, 128, True, 0),
    (2, 32, 1, 300, 128, True, 299), (2, 4, 100, 333, 128, True, 233), (1, 2, 130, 70, 80, False, 0),
]


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,q_pos0", CASES)
def test_attention_vs_torch(hip_lib, cuda, B, H, Sq, Sk, D, causal, q_pos0):
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + Sq + Sk + D)
    bf = torch.bfloat16
    q = torch.randn(B, H, Sq, D, generator=g).to(bf)
    k = torch.randn(B, H, Sk, D, generator=g).to(bf)
    v = torch.randn(B, H, Sk, D, generator=g).to(bf)
    scale = 1.0 / math.sqrt(D)
    ref = _ref(q, k, v, scale, causal, q_pos0)
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), scale, causal=causal, q_pos0=q_pos0)
    assert got.shape == (B, H, Sq, D)
    err = (got.float().cpu() - ref).abs().max().item()
    assert err < 2e-2, f"max err {err}"


def test_attention_fused_qkv_layout_and_kv_broadcast(hip_lib, cuda):
    """q/k/v read in place from a fused [B,S,3,H,D] qkv buffer (SAM layout); K/V broadcast over B."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(1)
    B, S, H, D = 2, 150, 4, 80
    qkv = torch.randn(B, S, 3, H, D, generator=g).to(torch.bfloat16).to(cuda)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    got = ops.attention(q, k, v, D ** -0.5)
    ref = _ref(q.cpu(), k.cpu(), v.cpu(), D ** -0.5)
    assert (got.float().cpu() - ref).abs().max().item() < 2e-2
    assert got.permute(0, 2, 1, 3).is_contiguous()  # [B,S,H,D] buffer feeds the proj GEMM directly
    # one K/V set shared by 4 query batches (SAM decoder image->token attention)
    q4 = torch.randn(4, H, 33, D, generator=g).to(torch.bfloat16).to(cuda)
    got = ops.attention(q4, k[:1], v[:1], 0.1)
    ref = _ref(q4.cpu(), k[:1].cpu().expand(4, -1, -1, -1), v[:1].cpu().expand(4, -1, -1, -1), 0.1)
    assert (got.float().cpu() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("SH,SW,B,H", [(14, 14, 3, 4), (64, 64, 1, 2)])
def test_relpos_attention(hip_lib, cuda, SH, SW, B, H, attn_block_shape):
    """SAM decomposed relative-position bias (image_encoder.py:321-392)."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(SH)
    D, S = 80, SH * SW
    bf = torch.bfloat16
    q = torch.randn(B, H, S, D, generator=g).to(bf)
    k = torch.randn(B, H, S, D, generator=g).to(bf)
    v = torch.randn(B, H, S, D, generator=g).to(bf)
    tab_h = (torch.randn(2 * SH - 1, D, generator=g) * 0.2).to(bf)
    tab_w = (torch.randn(2 * SW - 1, D, generator=g) * 0.2).to(bf)
    idx_h = torch.arange(SH)[:, None] - torch.arange(SH)[None, :] + (SH - 1)
    idx_w = torch.arange(SW)[:, None] - torch.arange(SW)[None, :] + (SW - 1)
    Rh, Rw = tab_h.float()[idx_h], tab_w.float()[idx_w]  # [SH,SH,D], [SW,SW,D]
    rq = q.float().reshape(B * H, SH, SW, D)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh).to(bf).float()  # reference rounds to the model dtype
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw).to(bf).float()
    bias = (rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).reshape(B, H, S, S)
    scale = D ** -0.5
    q_s = (q.float() * scale).to(bf)  # (q * self.scale) @ k^T in the model dtype (image_encoder.py:244)
    ref = _ref(q_s, k, v, 1.0, bias=bias)
    gh, gw = ops.relpos_bias(q.to(cuda), tab_h.to(cuda), tab_w.to(cuda), SH, SW)
    assert torch.allclose(gh.cpu(), rel_h.reshape(B * H, S, SH), atol=2e-2, rtol=1e-2)
    assert torch.allclose(gw.cpu(), rel_w.reshape(B * H, S, SW), atol=2e-2, rtol=1e-2)
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), scale, rel=(gh, gw))
    err = (got.float().cpu() - ref).abs().max().item()
    assert err < 3e-2, f"max err {err}"
    if SH == 64 and attn_block_shape == "4-wave":  # table mode of the global grid (REL 5): the terms computed in the kernel
        cat = ops.relpos_tables_cat(tab_h.to(cuda), tab_w.to(cuda))
        got_t = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), scale, rel_tab=(cat, SH))
        err_t = (got_t.float().cpu() - ref).abs().max().item()
        d = (got_t.float() - got.float()).abs().max().item()
        print(f"\n[64 x 64 grid, bf16] table mode vs reference {err_t:.2e} (array mode {err:.2e}), table vs array mode {d:.2e}")
        assert err_t < 3e-2 and d < 2e-2


def test_attention_rescale_branch_forced(hip_lib, cuda):
    """A late key with a huge score forces the online-softmax rescale of everything accumulated so far."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(4)
    bf = torch.bfloat16
    q = torch.randn(1, 2, 70, 64, generator=g).to(bf)
    k = torch.randn(1, 2, 400, 64, generator=g).to(bf)
    v = torch.randn(1, 2, 400, 64, generator=g).to(bf)
    k[0, :, 333] = (q[0, :, 7] * 6).to(bf)  # spike: key 333 aligned with query 7 (tile 5)
    ref = _ref(q, k, v, 0.125)
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), 0.125)
    assert (got.float().cpu() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("SH,SW,B,H", [(14, 14, 5, 4), (64, 64, 2, 3), (6, 10, 2, 2)])
def test_relpos_gemm_formulation_matches_dot_kernel(hip_lib, cuda, SH, SW, B, H):
    """rel-pos operands through one batched MFMA GEMM (q . [rel_pos_h ; rel_pos_w]^T, bf16 out) + the Toeplitz gather
    against the VALU dot-product kernel, on q read in place from a fused qkv buffer (SAM layout)."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(SH * 100 + SW)
    D, S = 80, SH * SW
    bf = torch.bfloat16
    qkv = torch.randn(B, S, 3, H, D, generator=g).to(bf).to(cuda)
    q = qkv[:, :, 0].permute(0, 2, 1, 3)
    tab_h = (torch.randn(2 * SH - 1, D, generator=g) * 0.2).to(bf).to(cuda)
    tab_w = (torch.randn(2 * SW - 1, D, generator=g) * 0.2).to(bf).to(cuda)
    eh, ew = ops.relpos_bias(q, tab_h, tab_w, SH, SW)                       # dot-product kernel
    cat = ops.relpos_tables_cat(tab_h, tab_w)
    assert cat.shape[0] % 8 == 0 and cat.shape[0] >= 2 * SH - 1 + 2 * SW - 1
    G = torch.empty(H, B * S, cat.shape[0], dtype=bf, device=cuda)          # the two steps by hand: every grid size
    from interactvlm_amd import _lib
    lib = _lib.load()
    ops.check(lib.ivlm_gemm_bf16(q.data_ptr(), q.stride(2), cat.data_ptr(), D, G.data_ptr(), cat.shape[0], 0, 0, 0, 0, B * S,
                                 cat.shape[0], D, 0, 0, H, q.stride(1), 0, B * S * cat.shape[0], 0, 0, 0.0, 0, 0, 0,
                                 torch.cuda.current_stream().cuda_stream), "gemm")
    gh, gw = torch.empty_like(eh), torch.empty_like(ew)
    ops.check(lib.ivlm_relpos_gather(G.data_ptr(), B * S * cat.shape[0], cat.shape[0], B, H, SH, SW, gh.data_ptr(), gw.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "gather")
    if min(SH, SW) >= 32:  # the path ops.relpos_bias takes by itself for large grids
        g2h, g2w = ops.relpos_bias(q, tab_h, tab_w, SH, SW, cat=cat)
        assert torch.equal(g2h, gh) and torch.equal(g2w, gw)
    assert gh.shape == eh.shape and gw.shape == ew.shape
    # both round an fp32 dot product of 80 terms to bf16: equal up to one bf16 ulp where the summation order flips a rounding
    for a, e in ((gh, eh), (gw, ew)):
        d = (a - e).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(e.abs().max()) + 1e-6
        assert float((d > 0).float().mean()) < 0.05
    # and against fp32 torch
    rq = q.float().reshape(B * H, SH, SW, D)
    idx_h = torch.arange(SH)[:, None] - torch.arange(SH)[None, :] + (SH - 1)
    idx_w = torch.arange(SW)[:, None] - torch.arange(SW)[None, :] + (SW - 1)
    Rh, Rw = tab_h.float()[idx_h.to(cuda)], tab_w.float()[idx_w.to(cuda)]
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh).reshape(B * H, S, SH)
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw).reshape(B * H, S, SW)
    assert torch.allclose(gh, ref_h, atol=2e-2, rtol=1e-2) and torch.allclose(gw, ref_w, atol=2e-2, rtol=1e-2)


# ---- IEEE fp16 operands (the default precision of the towers): the same kernels on v_mfma_f32_16x16x32_f16 ---------------------------
F16_CASES = [  # B, H, Sq, Sk, D, causal, q_pos0: the shapes of the path (CLIP, LLaMA prefill / chunked prefill)
    (1, 16, 257, 257, 64, False, 0), (2, 4, 70, 70, 64, False, 0), (1, 32, 330, 330, 128, True, 0), (2, 4, 100, 333, 128, True, 233),
]


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,q_pos0", F16_CASES)
def test_attention_f16_vs_torch(hip_lib, cuda, B, H, Sq, Sk, D, causal, q_pos0):
    """fp16 q / k / v: only the softmax weights and the output are rounded (to fp16): 8 x closer to the fp64 result than the bf16
    kernel on the same values (asserted)."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + Sq + Sk + D)
    q, k, v = (torch.randn(B, H, s_, D, generator=g).half() for s_ in (Sq, Sk, Sk))
    scale = 1.0 / math.sqrt(D)
    ref = _ref(q, k, v, scale, causal, q_pos0).double()  # (fp32 reference on the fp16 values: 1e-6)
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), scale, causal=causal, q_pos0=q_pos0, prescale_q=D == 64)
    assert got.dtype == torch.float16 and got.shape == (B, H, Sq, D)
    err = (got.double().cpu() - ref).abs().max().item()
    bf = ops.attention(q.bfloat16().to(cuda), k.bfloat16().to(cuda), v.bfloat16().to(cuda), scale, causal=causal, q_pos0=q_pos0,
                       prescale_q=D == 64)
    err_bf = (bf.double().cpu() - ref).abs().max().item()
    print(f"f16 attention max err {err:.2e} (bf16 operands of the same values: {err_bf:.2e})")
    assert err < 2.5e-3 and err < err_bf / 3, (err, err_bf)


@pytest.mark.parametrize("SH,SW,B,H", [(14, 14, 9, 4), (64, 64, 1, 2)])
def test_relpos_attention_f16(hip_lib, cuda, SH, SW, B, H):
    """SAM's attention with the decomposed rel-pos bias on fp16 operands: windows in table mode (the whole-window kernel and the
    generic flash kernel), the global grid through the fp16 GEMM formulation of the terms."""
    import torch

    from interactvlm_amd import _lib, ops

    g = torch.Generator().manual_seed(SH + 7)
    D, S = 80, SH * SW
    q, k, v = (torch.randn(B, S, H, D, generator=g).half().permute(0, 2, 1, 3) for _ in range(3))  # SAM's [B, S, H, D] rows
    tab_h = (torch.randn(2 * SH - 1, D, generator=g) * 0.2).bfloat16()
    tab_w = (torch.randn(2 * SW - 1, D, generator=g) * 0.2).bfloat16()
    idx_h = torch.arange(SH)[:, None] - torch.arange(SH)[None, :] + (SH - 1)
    idx_w = torch.arange(SW)[:, None] - torch.arange(SW)[None, :] + (SW - 1)
    Rh, Rw = tab_h.double()[idx_h], tab_w.double()[idx_w]
    rq = q.double().reshape(B * H, SH, SW, D)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    bias = (rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).reshape(B, H, S, S)
    scale = D ** -0.5
    sc = torch.einsum("bhqd,bhkd->bhqk", q.double() * scale, k.double()) + bias
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(sc, dim=-1), v.double())
    cat = ops.bf16_to_f16(ops.relpos_tables_cat(tab_h.to(cuda), tab_w.to(cuda)))
    qc, kc, vc = q.to(cuda), k.to(cuda), v.to(cuda)
    lib = _lib.load()
    if SH == 14:
        outs = []
        for v2 in (1, 0):
            lib.ivlm_attention_window_kernel(v2)
            outs.append(ops.attention(qc, kc, vc, scale, rel_tab=(cat, SH)))
        lib.ivlm_attention_window_kernel(1)
    else:
        gh, gw = ops.relpos_bias(qc, tab_h.to(cuda), tab_w.to(cuda), SH, SW, cat=cat)
        assert (gh.double().cpu() - rel_h.reshape(B * H, S, SH)).abs().max().item() < 2e-5  # (fp32 terms: exact products of fp16 q)
        assert (gw.double().cpu() - rel_w.reshape(B * H, S, SW)).abs().max().item() < 2e-5
        outs = [ops.attention(qc, kc, vc, scale, rel=(gh, gw)), ops.attention(qc, kc, vc, scale, rel_tab=(cat, SH))]  # arrays | table mode
    for o in outs:
        err = (o.double().cpu() - ref).abs().max().item()
        print(f"f16 rel-pos attention {SH}x{SW}: max err {err:.2e}")
        assert o.dtype == torch.float16 and err < 4e-3, err


@pytest.mark.parametrize("SH,SW,B,H", [(14, 14, 9, 4), (64, 64, 1, 2)])
def test_relpos_attention_f16_exact_q(hip_lib, cuda, SH, SW, B, H):
    """The "exact q" attention of the fp16 mode: q = hi + lo IEEE halves enters the rel-pos terms and the scores unrounded, the
    softmax weights are split for P.V - against fp64 on the fp32 q (k / v fp16): closer than the single-fp16-q kernel (asserted)."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(SH + 11)
    D, S = 80, SH * SW
    q32 = torch.randn(B, S, H, D, generator=g).permute(0, 2, 1, 3)
    k, v = (torch.randn(B, S, H, D, generator=g).half().permute(0, 2, 1, 3) for _ in range(2))
    tab_h = (torch.randn(2 * SH - 1, D, generator=g) * 0.5).bfloat16()
    tab_w = (torch.randn(2 * SW - 1, D, generator=g) * 0.5).bfloat16()
    idx_h = torch.arange(SH)[:, None] - torch.arange(SH)[None, :] + (SH - 1)
    idx_w = torch.arange(SW)[:, None] - torch.arange(SW)[None, :] + (SW - 1)
    rq = q32.double().reshape(B * H, SH, SW, D)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, tab_h.double()[idx_h])
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, tab_w.double()[idx_w])
    bias = (rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).reshape(B, H, S, S)
    scale = D ** -0.5
    sc = torch.einsum("bhqd,bhkd->bhqk", q32.double() * scale, k.double()) + bias
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(sc, dim=-1), v.double())
    q_hi = q32.half()
    q_lo = (q32 - q_hi.float()).half()
    cat = ops.bf16_to_f16(ops.relpos_tables_cat(tab_h.to(cuda), tab_w.to(cuda)))
    qh, ql, kc, vc = q_hi.to(cuda), q_lo.to(cuda), k.to(cuda), v.to(cuda)
    if SH == 14:
        o2 = ops.attention(qh, kc, vc, scale, rel_tab=(cat, SH), q_lo=ql, q_lo_level=2)
        o1 = ops.attention(qh, kc, vc, scale, rel_tab=(cat, SH), q_lo=ql, q_lo_level=1)
        o0 = ops.attention(qh, kc, vc, scale, rel_tab=(cat, SH))
    else:
        rel = ops.relpos_bias(qh, tab_h.to(cuda), tab_w.to(cuda), SH, SW, cat=cat, q_lo=ql)
        assert (rel[0].double().cpu() - rel_h.reshape(B * H, S, SH)).abs().max().item() < 3e-5
        o2 = ops.attention(qh, kc, vc, scale, rel=rel, q_lo=ql, q_lo_level=2)
        o1 = ops.attention(qh, kc, vc, scale, rel=rel)  # (level 1 with the terms as arrays IS the plain fp16 kernel)
        assert torch.equal(o1, ops.attention(qh, kc, vc, scale, rel=rel, q_lo=ql, q_lo_level=1))
        o0 = ops.attention(qh, kc, vc, scale, rel=ops.relpos_bias(qh, tab_h.to(cuda), tab_w.to(cuda), SH, SW, cat=cat))
        # table mode of the grid (REL 5) with the lo half of q in its table products == level 1 with the terms as arrays
        ot = ops.attention(qh, kc, vc, scale, rel_tab=(cat, SH), q_lo=ql, q_lo_level=1)
        dt = (ot.double() - o1.double()).abs().max().item()
        print(f"[64 x 64 grid, fp16 exact q] table mode vs array mode: {dt:.2e}")
        assert dt < 1e-3 and (ot.double().cpu() - ref).abs().max().item() < 2.5e-3
    e2, e1, e0 = ((o.double().cpu() - ref).abs().max().item() for o in (o2, o1, o0))
    print(f"exact-q fp16 attention {SH}x{SW}: max err level 2 {e2:.2e}, level 1 (rel-pos terms only) {e1:.2e}, single fp16 q {e0:.2e}")
    assert e2 < 1.5e-3 and e1 < 2.5e-3 and e2 <= e1 < e0, (e2, e1, e0)
