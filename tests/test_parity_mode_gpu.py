"""GPU tests of the "parity" precision mode: fp32-activation arithmetic on the bf16 matrix cores.
Every activation the default mode rounds to a bf16 MFMA operand travels as hi + lo bf16 halves (x = hi + lo to 2^-17):
GEMMs with a split A operand / split output, split-operand attention (three MFMAs per fragment), split norm outputs, split
RoPE + KV-cache planes.  References are torch fp64 on the SAME fp32 values (the kernels must be ~2^-17-accurate, where bf16
operand rounding gives 4e-3), then the three towers and the facade against the fp32 CPU oracle on identical weights."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16_weights(spec, seed=0):
    """fp32 weights whose values are exactly bf16-representable (shared by oracle and HIP path)."""
    import torch

    from interactvlm_amd import weights as Wt

    return {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(spec, seed).items()}


def _join(t_split, n):
    """[.., 2n] bf16 [hi | lo] -> fp32 [.., n]"""
    return t_split[..., :n].float() + t_split[..., n:].float()


def _rel(got, ref):
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max())


# (M, N, K): 128x64 tiles + split-K, 128x128 tiles, the 8-phase 256x256 kernel, a ragged edge
@pytest.mark.parametrize("M,N,K", [(330, 4096, 4096), (257, 1024, 1024), (2000, 1536, 1280), (16384, 3840, 1280),
                                   (4100, 1280, 5120), (300, 520, 192)])
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_gemm_split_operand_and_output(hip_lib, cuda, M, N, K, act):
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g)  # fp32 activations (NOT bf16-representable)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    if act == "gelu":
        ref = F.gelu(ref)
    ref = ref + r.double()
    xs = ops.split_rows(x.to(cuda))
    assert float((_join(xs, K).cpu() - x).abs().max()) < 2.0 ** -16 * float(x.abs().max())
    got = ops.linear(xs, w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda), out_f32=True, a_split=True)
    assert got.dtype == torch.float32 and got.shape == (M, N)
    e = _rel(got.cpu(), ref)
    assert e < 2e-5, f"a_split GEMM rel err {e}"
    # the same product with the bf16-rounded operand is ~100x further out: the test can tell the two apart
    e_bf = _rel(ops.linear(x.to(torch.bfloat16).to(cuda), w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda), out_f32=True).cpu(), ref)
    assert e_bf > 20 * e
    # split output == the fp32 output, split
    gs = ops.linear(xs, w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda), a_split=True, out_split=True)
    assert gs.dtype == torch.bfloat16 and gs.shape == (M, 2 * N)
    assert float((_join(gs, N) - got).abs().max()) <= 2.0 ** -16 * float(got.abs().max())
    assert torch.equal(gs[:, :N], got.to(torch.bfloat16))


def test_gemm_split_swiglu_rows_and_strides(hip_lib, cuda):
    """SwiGLU epilogue with split output (LLaMA gate|up), scatter / gather row maps (SAM window partition folded into the GEMMs)
    and strided output rows, all with a split A operand."""
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(5)
    M, K, I = 330, 1024, 1408
    x = torch.randn(M, K, generator=g)
    gu = (torch.randn(2 * I, K, generator=g) / K ** 0.5).to(torch.bfloat16)  # rows (gate_j, up_j) interleaved
    ref = F.silu(x.double() @ gu[0::2].double().T) * (x.double() @ gu[1::2].double().T)
    xs = ops.split_rows(x.to(cuda))
    got = ops.linear(xs, gu.to(cuda), act="swiglu", a_split=True, out_split=True)
    assert got.shape == (M, 2 * I)
    assert _rel(_join(got, I).cpu(), ref) < 3e-5
    # scatter epilogue + gather prologue
    M2, N2 = 2048, 768
    x2 = torch.randn(M2, K, generator=g)
    w2 = (torch.randn(N2, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    perm = torch.randperm(M2, generator=g).to(torch.int32)
    ref2 = x2.double() @ w2.double().T
    xs2 = ops.split_rows(x2.to(cuda))
    out = torch.zeros(M2, 2 * N2, dtype=torch.bfloat16, device=cuda)
    ops.linear(xs2, w2.to(cuda), out=out, out_rows=perm.to(cuda), a_split=True, out_split=True)
    assert _rel(_join(out, N2).cpu()[perm.long()], ref2) < 2e-5
    res = torch.randn(M2, N2, generator=g).to(cuda)
    got3 = ops.linear(xs2, w2.to(cuda), residual=res, out=res.clone(), a_rows=perm.to(cuda), a_split=True)
    assert _rel(got3.cpu(), ref2[perm.long()] + res.cpu().double()) < 2e-5


@pytest.mark.parametrize("cols", [256, 1280, 4096])
def test_norm_split_outputs(hip_lib, cuda, cols):
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(cols)
    x = (torch.randn(300, cols, generator=g) * 3).to(cuda)
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(torch.bfloat16).to(cuda)
    b = (0.1 * torch.randn(cols, generator=g)).to(torch.bfloat16).to(cuda)
    y32 = ops.layernorm(x, w, b, 1e-6, out_f32=True)
    ys = ops.layernorm(x, w, b, 1e-6, out_split=True)
    assert ys.shape == (300, 2 * cols) and float((_join(ys, cols) - y32).abs().max()) <= 2.0 ** -16 * float(y32.abs().max())
    r32 = ops.rmsnorm(x, w, 1e-5, out_f32=True)
    rs = ops.rmsnorm(x, w, 1e-5, out_split=True)
    assert float((_join(rs, cols) - r32).abs().max()) <= 2.0 ** -16 * float(r32.abs().max())


def _attn_ref(q, k, v, scale, causal=False, q_pos0=0, bias=None):
    import torch

    s = torch.einsum("bhqd,bhkd->bhqk", q.double() * scale, k.double())
    if bias is not None:
        s = s + bias.double()
    if causal:
        Sq, Sk = q.shape[2], k.shape[2]
        s = s.masked_fill(torch.arange(Sk)[None, :] > torch.arange(Sq)[:, None] + q_pos0, float("-inf"))
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, dim=-1), v.double())


def _split_planes(t, cuda):
    """fp32 [B,H,S,D] -> (hi, lo) bf16 device tensors laid out like the GEMM's split output ([B,S,2,H,D] buffer)."""
    import torch

    B, H, S, D = t.shape
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    buf = torch.stack([hi.permute(0, 2, 1, 3), lo.permute(0, 2, 1, 3)], 2).contiguous().to(cuda)  # [B,S,2,H,D]
    return buf[:, :, 0].permute(0, 2, 1, 3), buf[:, :, 1].permute(0, 2, 1, 3)


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,q_pos0", [(2, 16, 257, 257, 64, False, 0), (1, 8, 330, 330, 128, True, 0),
                                                       (1, 4, 100, 333, 128, True, 233), (2, 4, 70, 70, 128, True, 0)])
def test_attention_split_vs_fp64(hip_lib, cuda, B, H, Sq, Sk, D, causal, q_pos0):
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(Sq + Sk + D)
    q, k, v = (torch.randn(B, H, s, D, generator=g) for s in (Sq, Sk, Sk))
    scale = 1.0 / math.sqrt(D)
    ref = _attn_ref(q, k, v, scale, causal, q_pos0)
    (qh, ql), (kh, kl), (vh, vl) = (_split_planes(t, cuda) for t in (q, k, v))
    for pre in ((True, False) if not causal else (False,)):
        out = ops.attention_split(qh, ql, kh, kl, vh, vl, scale, causal=causal, q_pos0=q_pos0, prescale_q=pre)
        got = _join(out, H * D).view(B, Sq, H, D).permute(0, 2, 1, 3).cpu()
        e = float((got.double() - ref).abs().max())
        assert e < 3e-5, f"split attention max err {e}"
    # the bf16 kernel on the rounded operands is two orders of magnitude further out
    o16 = ops.attention(qh.contiguous(), kh.contiguous(), vh.contiguous(), scale, causal=causal, q_pos0=q_pos0)
    assert float((o16.float().cpu().double() - ref).abs().max()) > 20 * e


@pytest.mark.parametrize("B,side", [(6, 14), (1, 64)])
def test_sam_attention_split_with_relpos(hip_lib, cuda, B, side):
    """SAM ViT-H attention (head dim 80) with the decomposed rel-pos bias: 14x14 windows (bias folded into the QK^T MFMA as a
    one-hot product, hi + lo) and the 64x64 global grid (bias from registers / LDS), q scaled before the product."""
    import torch

    from interactvlm_amd import ops

    H, D, S = 4, 80, side * side
    g = torch.Generator().manual_seed(side)
    q, k, v = (torch.randn(B, H, S, D, generator=g) for _ in range(3))
    th = (0.5 * torch.randn(2 * side - 1, D, generator=g)).to(torch.bfloat16)
    tw = (0.5 * torch.randn(2 * side - 1, D, generator=g)).to(torch.bfloat16)
    idx = torch.arange(side)[:, None] - torch.arange(side)[None, :] + side - 1
    rq = q.double().view(B, H, side, side, D)
    rel_h = torch.einsum("bhyxd,ykd->bhyxk", rq, th.double()[idx])  # [B,H,qy,qx,ky]
    rel_w = torch.einsum("bhyxd,xkd->bhyxk", rq, tw.double()[idx])
    bias = (rel_h[..., :, None] + rel_w[..., None, :]).reshape(B, H, S, S)
    scale = D ** -0.5
    ref = _attn_ref(q, k, v, scale, bias=bias)
    (qh, ql), (kh, kl), (vh, vl) = (_split_planes(t, cuda) for t in (q, k, v))
    rh, rw = ops.relpos_bias_split(qh, ql, th.to(cuda), tw.to(cuda), side, side)
    assert float((rh.cpu().double() - rel_h.reshape(B * H, S, side)).abs().max()) < 1e-4 * float(rel_h.abs().max())
    assert float((rw.cpu().double() - rel_w.reshape(B * H, S, side)).abs().max()) < 1e-4 * float(rel_w.abs().max())
    out = ops.attention_split(qh, ql, kh, kl, vh, vl, scale, rel=(rh, rw))
    got = _join(out, H * D).view(B, S, H, D).permute(0, 2, 1, 3).cpu()
    e = float((got.double() - ref).abs().max())
    # the floor is the 2^-17 relative split of v (|v| up to 4.5) under a peaked softmax (biases of +-20): ~7e-5; bf16 operands: 1e-2
    assert e < 1.5e-4, f"SAM split attention ({side}x{side}) max err {e}"
    o16 = ops.attention(qh.contiguous(), kh.contiguous(), vh.contiguous(), scale, rel=(rh, rw))
    assert float((o16.float().cpu().double() - ref).abs().max()) > 20 * e
    if 2 * side <= 32:  # TABLE MODE: the kernel computes the rel-pos terms itself from [rel_pos_h ; rel_pos_w]
        tab = ops.relpos_table64(th.to(cuda), tw.to(cuda))
        out_t = ops.attention_split(qh, ql, kh, kl, vh, vl, scale, rel_tab=(tab, side))
        got_t = _join(out_t, H * D).view(B, S, H, D).permute(0, 2, 1, 3).cpu()
        e_t = float((got_t.double() - ref).abs().max())
        assert e_t < 1.5e-4, f"table-mode split attention max err {e_t}"
        # default precision: same numbers as the relpos kernel + array mode on the bf16 q (terms rounded to bf16 in both)
        qb, kb, vb = qh.contiguous(), kh.contiguous(), vh.contiguous()
        a_arr = ops.attention(qb, kb, vb, scale, rel=ops.relpos_bias(qb, th.to(cuda), tw.to(cuda), side, side))
        a_tab = ops.attention(qb, kb, vb, scale, rel_tab=(tab, side))
        d = float((a_arr.float() - a_tab.float()).abs().max())
        assert d < 2e-2, d  # (fp32 summation order of the table product may flip a bf16 rounding of a term)


def test_rope_split_cache_and_decode_attention(hip_lib, cuda):
    """RoPE on split q|k|v rows + append to hi + lo cache planes, then the single-token and the batched decode attention
    reading those planes, against fp64."""
    import torch

    from interactvlm_amd import ops

    H, D, T, Tmax = 4, 128, 37, 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(T + 1, 3 * H * D, generator=g)
    cos, sin = ops.rope_table(Tmax, D, 10000.0, cuda)
    caches = [torch.zeros(Tmax, H, D, dtype=torch.bfloat16, device=cuda) for _ in range(4)]  # k, k_lo, v, v_lo
    qs = ops.split_rows(qkv[:T].to(cuda))
    ops.rope_kv_split(qs, H, D, 0, tuple(caches), (cos, sin))

    def rope(x, pos):  # x [T,H,D] fp64
        half = D // 2
        c, s = cos.cpu().double()[pos][:, None, :], sin.cpu().double()[pos][:, None, :]
        return torch.cat([x[..., :half] * c - x[..., half:] * s, x[..., half:] * c + x[..., :half] * s], -1)

    q3 = qkv.double().view(T + 1, 3, H, D)
    pos = torch.arange(T + 1)
    qr, kr = rope(q3[:, 0], pos), rope(q3[:, 1], pos)
    got_q = _join(qs, 3 * H * D).cpu().view(T, 3, H, D)[:, 0]
    assert float((got_q.double() - qr[:T]).abs().max()) < 1e-4
    kc = caches[0].float() + caches[1].float()
    vc = caches[2].float() + caches[3].float()
    assert float((kc[:T].cpu().double() - kr[:T]).abs().max()) < 1e-4
    assert float((vc[:T].cpu().double() - q3[:T, 2]).abs().max()) < 1e-4
    # one decode step at position T on the split cache
    o = ops.llama_decode_attn(qkv[T: T + 1].to(cuda).contiguous(), caches[0], caches[2], H, D, T, 10000.0, D ** -0.5,
                              table=(cos, sin), lo=(caches[1], caches[3]))
    s = torch.einsum("hd,thd->ht", qr[T], kr) * D ** -0.5
    ref = torch.einsum("ht,thd->hd", torch.softmax(s, -1), q3[:, 2]).reshape(1, H * D)
    assert float((o.cpu().double() - ref).abs().max()) < 2e-5
    kc2 = caches[0].float() + caches[1].float()
    assert float((kc2[T].cpu().double() - kr[T]).abs().max()) < 1e-4  # the new row was appended unrounded
    # batched: two sequences sharing the layout [B, Tmax, H, D]
    bc = [torch.stack([c, c]).contiguous() for c in caches]
    for c in bc:
        c[:, T] = 0
    pos_dev = torch.tensor([T, T], dtype=torch.int32, device=cuda)
    ob = ops.llama_decode_attn_batch(qkv[T: T + 1].repeat(2, 1).to(cuda).contiguous(), bc[0], bc[2], H, D, pos_dev, 10000.0,
                                     D ** -0.5, table=(cos, sin), lo=(bc[1], bc[3]))
    assert float((ob.cpu().double() - ref).abs().max()) < 2e-5


def test_towers_parity_mode_vs_oracle(hip_lib, cuda):
    """SAM ViT (real width, 4 blocks incl. a global one), CLIP (4 layers) and LLaMA (3 layers, prefill + decode) in parity
    mode against the fp32 CPU oracle on identical bf16-valued weights: relative errors at the fp32 level (the default mode sits
    at 1e-2)."""
    import torch

    from interactvlm_amd import llava, sam
    from interactvlm_amd import weights as Wt
    from oracle import nn as O

    torch.set_grad_enabled(False)
    rel = lambda a, b: float((a.float().cpu() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    c = Wt.SamEncCfg(depth=4, global_attn_indexes=(2,))
    w = _bf16_weights(Wt.sam_encoder_spec(c))
    enc = sam.SamImageEncoder(w, c, cuda)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 1024, 1024, generator=g).to(torch.bfloat16)
    ref = O.sam_image_encoder(w, Wt.SAM_PREFIX + ".image_encoder", x.float(), c.depth, c.num_heads, c.global_attn_indexes)
    ref = ref.permute(0, 2, 3, 1).reshape(2, 4096, 256)
    e_def = rel(enc(x.to(cuda)), ref)
    enc.precision = "parity"
    e_par = rel(enc(x.to(cuda)), ref)
    e_par2 = rel(enc(x.to(cuda)), ref)  # graph replay
    # fp16 operands (the model's default mode): plain, and with the exact q path
    enc.parity_sites = enc.SITES_F16
    e_f16 = rel(enc(x.to(cuda)), ref)
    enc.parity_sites = enc.SITES_F16Q
    e_f16q = rel(enc(x.to(cuda)), ref)
    enc.q_lo_level = 2
    e_f16q2 = rel(enc(x.to(cuda)), ref)
    print(f"\n[SAM ViT-H width, 4 blocks] rel rms err: bf16 operands {e_def:.2e}, fp16 {e_f16:.2e}, fp16 + exact q {e_f16q:.2e} "
          f"(lo half of q in Q.K^T too: {e_f16q2:.2e}), parity {e_par:.2e}")
    assert e_par < 1e-4 and e_par2 < 1e-4 and e_def > 10 * e_par
    assert e_f16 < e_def / 5 and e_f16q < e_f16 and e_f16q2 < 1.05 * e_f16q
    del enc

    cc = Wt.ClipCfg(hidden=256, layers=4, heads=4, inter=512)
    w = _bf16_weights(Wt.clip_spec(cc))
    xc = torch.randn(2, 3, 224, 224, generator=g).to(torch.bfloat16)
    tower = llava.ClipTower(w, cc, cuda)
    refc = O.clip_vision(w, Wt.CLIP_PREFIX, xc.float(), 4, 4)
    e_def = rel(tower(xc.to(cuda)), refc)
    tower.precision = "parity"
    fs = tower(xc.to(cuda))
    assert fs.shape == (2, 256, 512)
    e_par = rel(_join(fs, 256), refc)
    tower.precision = "f16"
    e_f16 = rel(_join(tower(xc.to(cuda)), 256), refc)
    print(f"[CLIP, 3 layers run] rel rms err: bf16 operands {e_def:.2e}, fp16 {e_f16:.2e}, parity {e_par:.2e}")
    assert e_par < 5e-5 and e_def > 10 * e_par and e_f16 < e_def / 5

    lc = Wt.LlamaCfg(hidden=512, layers=3, heads=4, inter=1024, vocab=1000)
    w = _bf16_weights(Wt.llama_spec(lc))
    llm = llava.Llama(w, lc, cuda, max_len=256)
    emb = (torch.randn(90, 512, generator=g) * 0.5).to(torch.bfloat16).float()
    refl = O.llama(w, "model", emb[None], lc.layers, lc.heads)[0]
    errs = {}
    for mode in ("default", "parity", "f16"):
        llm.set_precision(mode)
        h = [llm.forward(emb[:70].to(cuda), 0)]
        for t in range(70, 90):
            h.append(llm.forward(emb[t: t + 1].to(cuda), t))
        got = torch.cat(h, 0)
        errs[mode] = (rel(got[:70], refl[:70]), rel(got[70:], refl[70:]))
    print(f"[LLaMA, 3 layers] rel rms err (prefill rows, decode rows): bf16 operands {errs['default']}, fp16 operands + fp16 KV cache "
          f"{errs['f16']}, parity {errs['parity']}")
    assert max(errs["parity"]) < 5e-5 and errs["default"][0] > 10 * errs["parity"][0]
    assert errs["f16"][0] < errs["default"][0] / 5 and errs["f16"][1] < errs["default"][1] / 5
    # a short chunk (2 .. 16 rows) in the non-bf16 modes goes token by token through the fp32-activation decode kernels
    for mode in ("parity", "f16"):
        llm.set_precision(mode)
        h = torch.cat([llm.forward(emb[:70].to(cuda), 0), llm.forward(emb[70:78].to(cuda), 70)], 0)
        assert rel(h, refl[:78]) < 2 * max(errs[mode]) + 1e-5


def test_evaluate_parity_mode_vs_oracle(hip_lib, cuda):
    """The facade in parity mode on the structurally complete tiny configuration: evaluate() and evaluate_batch() against the
    fp32 oracle on identical weights - per-vertex contacts far inside the 1e-3 target, threshold sets equal."""
    import torch

    from interactvlm_amd import model as M
    from interactvlm_amd import synth, synthetic
    from interactvlm_amd import weights as Wt
    from oracle import pipeline as P

    torch.set_grad_enabled(False)
    cfg = synthetic.config_tiny()
    w = {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(Wt.ivlm_spec(cfg)).items()}
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, cuda, lift_tables=tables)
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=8)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, cuda)
    full_ids = torch.cat([ids[0], torch.tensor(forced)])
    ref = P.model_forward(w, cfg, im[0].float().cpu(), ic.float().cpu(), full_ids, cams[0], tables)["pred_contact"]
    errs = {}
    for mode in ("default", "bf16", "parity", "default"):
        m.set_precision(mode)
        out = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)
        errs[mode] = float((out["pred_contact_3d"].float().cpu() - ref).abs().max())
    print(f"\n[evaluate, tiny] max |dp| vs fp32 oracle: default (fp16 operands) {errs['default']:.2e}, bf16 {errs['bf16']:.2e}, "
          f"parity {errs['parity']:.2e}")
    assert errs["parity"] < 1e-4 and errs["default"] < 2e-4 and errs["bf16"] < 1e-3
    m.set_precision("parity")
    single = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)["pred_contact_3d"]
    ic2, im2 = torch.cat([ic, ic]), torch.cat([im, im])
    outs = m.evaluate_batch(ic2, im2, [ids[0], ids[0]], [cams[0], cams[0]], [(1024, 1024)] * 2, [(1024, 1024)] * 2,
                            forced_new_tokens=forced)
    for o in outs:
        assert float((o["pred_contact_3d"] - single).abs().max()) < 1e-4
        assert float((o["pred_contact_3d"].float().cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("seed,oracle", [(3, True), (12, False), (24, False)])
def test_full_depth_end_to_end_vs_oracle(hip_lib, cuda, seed, oracle):
    """End-to-end parity at the REAL depths: SAM ViT-H with all 32 blocks at its real width on the four views,
    a 32-layer LLaMA (width 1024: the fp32 oracle of the 7B width needs 27 GB of host weights - bench.py's
    `parity_vs_oracle_full_depth` leg does exactly that on the headline model), a 23-layer CLIP, the real mask decoder, 4 x 1024^2
    masks and the 6890-vertex lift: evaluate() against the fp32 CPU oracle on identical bf16-valued weights.
    EVERY listed mode except "bf16" must hold the north star's 1e-3 on per-vertex probabilities: the DEFAULT mode (fp16 operands,
    exact q path: 4.1 - 6.7e-4 over 16 seeded sets, profiles/r04_default_mode_seeds.txt), "parity" / "parity-fast" with exactly
    equal vertex-id sets; the bf16-operand mode's error at this depth (the reference's own GPU dtype class) is printed and
    sanity-bounded.
    THREE weight / image seeds (the bound is the north star's 1e-3 itself over several seeds - among them the
    family's worst, 12 and 24 - not one seed at a hand-tuned 7e-4).  Seed 3 runs the fp32 CPU oracle (~100 s); for the other two the
    reference is this build's "parity" mode, which seed 3 pins to < 2e-5 of the oracle, with that distance taken off the bound."""
    import time

    import torch

    from interactvlm_amd import model as M
    from interactvlm_amd import ops, synthetic
    from interactvlm_amd import weights as Wt
    from oracle import cref
    from oracle import nn as O
    from oracle import pipeline as P

    torch.set_grad_enabled(False)
    cfg = Wt.IvlmCfg(llama=Wt.LlamaCfg(hidden=1024, layers=32, heads=8, inter=2752, vocab=32003),
                     clip=Wt.ClipCfg(hidden=256, layers=24, heads=4, inter=512), sam=Wt.SamEncCfg())
    wd = synthetic.device_weights(cfg, cuda, seed=seed)
    tables = synthetic.body_lift_tables(cuda)
    m = M.InteractVLMForCausalLM(cfg, wd, cuda, lift_tables=tables)
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, cuda, seed=seed + 2)
    got = {}
    for mode in (m.precision_modes if oracle else ("default", "parity")):
        m.set_precision(mode)
        o = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)
        plan = m.human_3d_contact_predictor._get_plan(cuda)
        _, nv = ops.lift_mesh_plan(o["pred_masks"][0][None].contiguous(), plan, want_nviews=True)
        got[mode] = (o["pred_contact_3d"].float().cpu(), nv[0].cpu() > 0)
    del m
    if not oracle:  # reference = the parity mode (pinned to the oracle by the seed-3 case below: < 2e-5)
        e = float((got["default"][0] - got["parity"][0]).abs().max())
        print(f"\n[full depth, seed {seed}] default mode vs the parity mode: max |dp| = {e:.2e}")
        assert e < 1e-3 - 2e-5
        assert torch.equal(got["default"][1], got["parity"][1])  # visibility set: bit-exact
        for thr, op in ((0.5, torch.ge), (0.3, torch.gt)):  # threshold sets of the default mode against the parity mode's: off-band equal
            same = op(got["default"][0], thr) == op(got["parity"][0], thr)
            band = (got["parity"][0] - thr).abs() <= e + 2e-5
            assert bool(same[~band].all())
            print(f"[full depth, seed {seed}] default: {int((~same).sum())} of {int(band.sum())} band vertices flip at {thr} (vs parity mode)")
        return
    t0 = time.time()
    w = {k: v.float().cpu() for k, v in wd.items()}
    full_ids = torch.cat([ids[0], torch.tensor(forced)])
    feat = P.encode_images(w, cfg, ic.float().cpu())[0]
    hidden = P.llm_hidden(w, cfg, full_ids, feat)
    rows = O.seg_rows(full_ids, [cfg.seg_token_idx], cfg.img_emb_len, model_forward=True)
    seg_emb = O.text_hidden_fcs(w, hidden)[rows]
    token = int(full_ids[int(rows.nonzero()[0]) - cfg.img_emb_len + 1])
    imc = im[0].float().cpu()
    emb = torch.cat([P.sam_embed(w, cfg, imc[v: v + 1]) for v in range(imc.shape[0])], 0)  # one view at a time (host memory)
    masks, _, _ = P.decode_masks(w, cfg, seg_emb, token, cams[0], emb, (1024, 1024), (1024, 1024))
    ref, nviews = cref.lift_mesh_soft(masks.numpy()[None], tables[0].cpu().numpy().astype(np.int32), tables[1].cpu().numpy(), 6890)
    ref = torch.from_numpy(ref)
    err = {mode: float((c - ref).abs().max()) for mode, (c, _) in got.items()}
    print(f"\n[full depth: SAM ViT-H 32 blocks x 4 views, 32-layer LLaMA, 23-layer CLIP] max |dp| vs fp32 oracle: default (fp16 "
          f"operands, exact q) {err['default']:.2e}, bf16 {err['bf16']:.2e}, parity-fast {err['parity-fast']:.2e}, parity "
          f"{err['parity']:.2e} (oracle {time.time() - t0:.0f} s)")
    c, vis = got["parity"]
    assert err["default"] < 1e-3  # the mode `value` of bench.py is quoted on: the north star's bound itself (see the docstring)
    assert err["parity"] < 2e-5 and err["parity-fast"] < 1e-3  # ("parity" is the other seeds' reference)
    assert torch.equal(got["default"][1], vis) and torch.equal(got["parity-fast"][1], vis)
    assert torch.equal(vis, torch.from_numpy(nviews[0] > 0))  # visibility set: bit-exact (every mode)
    # The north star's "vertex-id sets bit-exact" ({p >= 0.5} eval_utils.py:75, {p > 0.3} run_demo.py:459), per mode
    # [r6, VERDICT r5 item 4]: the `parity` mode's sets EQUAL the oracle's, exactly; the default mode's are equal outside its error band
    # and the number of band vertices that land on the other side is printed (bench.py: value_sets_exactly_equal / value_sets_detail).
    for mode in ("parity", "parity-fast", "default"):
        cm = got[mode][0]
        for thr, op in ((0.5, torch.ge), (0.3, torch.gt)):
            same = op(cm, thr) == op(ref, thr)
            band = (ref - thr).abs() <= err[mode]
            assert bool(same[~band].all())
            if mode == "parity":
                assert bool(same.all()), f"parity mode: {int((~same).sum())} vertices flip at {thr}"
            else:
                print(f"[full depth, seed {seed}] {mode}: {int((~same).sum())} of {int(band.sum())} band vertices flip at {thr}")
    assert err["bf16"] < 5e-2


@pytest.mark.parametrize("M,N,K,act", [(16384, 5120, 1280, "gelu"), (16384, 1280, 5120, "none"), (300, 512, 256, "none")])
def test_gemm_fp16_operands_and_output(hip_lib, cuda, M, N, K, act):
    """IEEE-half operands on the tile GEMMs (the MLP of the SAM encoder in 'parity-encoder' precision): exact products of the fp16
    values, fp32 accumulation, fp16 / fp32 outputs, fp32 residual."""
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.float16)
    wb = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    w = wb.to(torch.float16)  # exact above the fp16 subnormal range, off by <= 2^-25 below it
    assert float((w.float() - wb.float()).abs().max()) <= 2.0 ** -25
    assert torch.equal(ops.bf16_to_f16(wb.to(cuda)).cpu(), w)  # ivlm_bf16_to_f16: RNE like torch's conversion
    # out of range -> inf, like torch's conversion (NOT clamped: an overflow must stay visible, DESIGN.md par. 3); odd length (tail path)
    big = torch.tensor([1e6, -3e38, 65504.0, 7.0, 1e-9, -0.0, 3.0], dtype=torch.bfloat16)
    assert torch.equal(ops.bf16_to_f16(big.to(cuda)).cpu(), big.to(torch.float16))
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    if act == "gelu":
        ref = F.gelu(ref)
    got = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda), out_f32=True)
    assert _rel(got.cpu(), ref + r.double()) < 2e-5
    g16 = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act=act, out_f16=True)
    assert g16.dtype == torch.float16 and _rel(g16.float().cpu(), ref) < 1e-3  # one fp16 rounding of the result
    # LayerNorm with an fp16 output == its fp32 output rounded to fp16
    xf = (torch.randn(64, K, generator=g) * 2).to(cuda)
    lw, lb = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).to(cuda), (0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).to(cuda)
    assert torch.equal(ops.layernorm(xf, lw, lb, 1e-6, out_f16=True), ops.layernorm(xf, lw, lb, 1e-6, out_f32=True).to(torch.float16))


def test_headline_model_7b_default_mode(hip_lib, cuda):
    """the HEADLINE model at its real widths under pytest - LLaMA 32 x 4096 (inter 11008, vocab 32003), CLIP ViT-L/14
    (1024), SAM ViT-H, 4 x 1024^2 views, the 6890-vertex lift (`synthetic.config_7b()`, BASELINE configs[1]; reference path
    model/InteractVLM.py:510-638).  The fp32 CPU oracle of this model takes ~110 s and 27 GB (bench.py's parity leg runs it); here,
    without it: (i) the DEFAULT mode's result is finite; (ii) default vs the "parity" mode < 1e-3 (parity sits 8e-6 from the oracle
    on this very model - BENCH `parity_vs_oracle_full_depth` - so this is the oracle check by proxy); (iii) evaluate_batch(2) ==
    evaluate per image < 1e-3, ids equal; (iv) the lift of its own masks through the C oracle (`cref.lift_mesh_soft`): contacts
    1e-5, the visibility set bit-exact; (v) the packed decode step vs the bf16-weight step (`decode_packed = False`) on the hidden
    states - same exact products, other fp32 summation order (matrix-core accumulation vs wave reduction), so equal to rounding, not
    bit for bit; (vi) free-running greedy generation (no forced ids, no per-token host sync) == the eager loop, token for token;
    (vii) the default mode holds no bf16 copy of a packed LLaMA matrix (resident weight bytes by form)."""
    import numpy as np
    import torch

    from interactvlm_amd import model as M
    from interactvlm_amd import ops, synthetic
    from oracle import cref

    torch.set_grad_enabled(False)
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, cuda, seed=0)
    tables = synthetic.body_lift_tables(cuda)
    m = M.InteractVLMForCausalLM(cfg, w, cuda, lift_tables=tables)
    del w
    torch.cuda.empty_cache()
    assert m.precision == "default"
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, cuda, seed=0)
    S = cfg.sam.img_size
    ev = lambda: m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)
    # (vii) residency: fp16 prefill copies + 12-bit decode planes, no bf16 originals of the packed matrices
    rb = m.resident_weight_bytes()
    assert m.llm.prefill_panels and all("qkv_h" not in L for L in m.llm.layers)  # (fp16 prefill copies in the K-panel layout only)
    n_mat = sum(L[n + "_hp"].numel() for L in m.llm.layers for n in ("qkv", "o", "gu", "down"))
    assert rb["f16"] == 2 * n_mat and 1.45 * n_mat < rb["bf12"] < 1.6 * n_mat + 1.6 * m.llm.lm_head.numel()
    assert rb["bf16"] < 2.1 * (m.llm.embed.numel() + m.llm.lm_head.numel())  # embed_tokens + lm_head + norms only
    # (i) default mode
    out = ev()
    pc = out["pred_contact_3d"].float().cpu()
    assert "recomputed_in_bf16" not in out and pc.shape == (1, 6890) and bool(torch.isfinite(pc).all())
    assert 0.0 <= float(pc.min()) and float(pc.max()) <= 1.0 and float(pc.std()) > 1e-3  # (not a constant map)
    # (iv) the lift of its own masks, C oracle
    masks = out["pred_masks"][0]
    assert masks.shape == (4, S, S) and masks.dtype == torch.float32
    ref, nviews = cref.lift_mesh_soft(masks.cpu().numpy()[None], tables[0].cpu().numpy().astype(np.int32), tables[1].cpu().numpy(), 6890)
    assert float((pc - torch.from_numpy(ref)).abs().max()) < 1e-5
    plan = m.human_3d_contact_predictor._get_plan(cuda)
    _, nv = ops.lift_mesh_plan(masks[None].contiguous(), plan, want_nviews=True)
    assert torch.equal(nv[0].cpu() > 0, torch.from_numpy(nviews[0] > 0))
    # (iii) two images per call
    ic2, im2 = synthetic.images(cfg, cuda, seed=1)
    outs = m.evaluate_batch(torch.cat([ic, ic2]), torch.cat([im, im2]), [ids[0], ids[0]], [cams[0], cams[0]], [(S, S)] * 2, [(S, S)] * 2,
                            forced_new_tokens=forced)
    single2 = m.evaluate(ic2, im2, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)
    for o, one in zip(outs, (out, single2)):
        assert torch.equal(o["output_ids"], one["output_ids"])
        e = float((o["pred_contact_3d"] - one["pred_contact_3d"]).abs().max())
        assert e < 1e-3, e
    # (v) packed decode weights vs bf16 decode weights: hidden states of the whole forced generation
    _, h_packed = m.generate(ic, ids, forced_new_tokens=forced)
    m.llm.decode_packed = False
    _, h_bf16 = m.generate(ic, ids, forced_new_tokens=forced)
    m.llm.decode_packed = True
    T0 = ids.shape[1] - 1 + cfg.img_emb_len + 1
    assert torch.equal(h_packed[:T0], h_bf16[:T0])  # (the prefill does not depend on the decode weights' form)
    d = (h_packed[T0:] - h_bf16[T0:]).abs().max() / h_bf16[T0:].abs().max()
    assert float(d) < 2e-4, float(d)  # (fp32 summation order only, compounded over 32 layers x 23 steps; a lost bit of a weight: > 1e-2)
    # (vi) free-running greedy search: graph replay without per-token host sync == eager loop
    n_new = 12
    m.graph_decode = True
    g_ids, g_hidden = m.generate(ic, ids, max_new_tokens=n_new, eos_token_id=2)
    m.graph_decode = False
    e_ids, e_hidden = m.generate(ic, ids, max_new_tokens=n_new, eos_token_id=2)
    m.graph_decode = True
    assert torch.equal(g_ids, e_ids) and torch.equal(g_hidden, e_hidden)
    # (ii) the oracle by proxy: parity mode (bf16 originals rebuilt, bit for bit, from the packed planes)
    m.set_precision("parity")
    rb2 = m.resident_weight_bytes()
    assert rb2["f16"] == 0 and rb2["bf16"] >= 2 * n_mat
    par = ev()["pred_contact_3d"].float().cpu()
    e = float((pc - par).abs().max())
    print(f"\n[7B headline model] default mode vs parity mode: max |dp| = {e:.2e}; resident LLaMA weights default mode "
          f"{sum(rb.values()) / 1e9:.1f} GB (bf16 {rb['bf16'] / 1e9:.1f}, fp16 {rb['f16'] / 1e9:.1f}, 12-bit {rb['bf12'] / 1e9:.1f})")
    assert e < 1e-3 - 2e-5


def test_headline_model_13b_default_mode(hip_lib, cuda):
    """the RELEASED checkpoints' size (LLaVA-1.5-13B: 40 x 5120, inter 13824, 40 heads - run_demo.py:33,
    scripts/run_train.sh:58; `synthetic.config_13b()`) under pytest, the checks (i) - (iv) and (vii) of the 7B test above: (i) the
    DEFAULT mode's result is finite and not constant; (ii) default vs the "parity" mode < 1e-3 (the oracle by proxy - `bench.py --model
    13b` runs the fp32 CPU oracle itself: profiles/r06_bench_13b.json `parity_vs_oracle_full_depth`), threshold sets equal off the
    error band, visibility set bit-exact; (iii) evaluate_batch(2) == evaluate per image; (iv) the lift of its own masks through the C
    oracle; (vii) residency by form (no bf16 copy of a packed matrix in the default mode)."""
    import numpy as np
    import torch

    from interactvlm_amd import model as M
    from interactvlm_amd import ops, synthetic
    from oracle import cref

    torch.set_grad_enabled(False)
    cfg = synthetic.config_13b()
    w = synthetic.device_weights(cfg, cuda, seed=0)
    tables = synthetic.body_lift_tables(cuda)
    m = M.InteractVLMForCausalLM(cfg, w, cuda, lift_tables=tables)
    del w
    torch.cuda.empty_cache()
    assert m.precision == "default" and len(m.llm.layers) == 40
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, cuda, seed=0)
    S = cfg.sam.img_size
    ev = lambda: m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)
    rb = m.resident_weight_bytes()  # (vii)
    n_mat = sum(L[n + "_hp"].numel() for L in m.llm.layers for n in ("qkv", "o", "gu", "down"))
    assert rb["f16"] == 2 * n_mat and 1.45 * n_mat < rb["bf12"] < 1.6 * n_mat + 1.6 * m.llm.lm_head.numel()
    assert rb["bf16"] < 2.1 * (m.llm.embed.numel() + m.llm.lm_head.numel())
    out = ev()  # (i)
    pc = out["pred_contact_3d"].float().cpu()
    assert "recomputed_in_bf16" not in out and pc.shape == (1, 6890) and bool(torch.isfinite(pc).all())
    assert 0.0 <= float(pc.min()) and float(pc.max()) <= 1.0 and float(pc.std()) > 1e-3
    masks = out["pred_masks"][0]  # (iv)
    ref, nviews = cref.lift_mesh_soft(masks.cpu().numpy()[None], tables[0].cpu().numpy().astype(np.int32), tables[1].cpu().numpy(), 6890)
    assert float((pc - torch.from_numpy(ref)).abs().max()) < 1e-5
    plan = m.human_3d_contact_predictor._get_plan(cuda)
    _, nv = ops.lift_mesh_plan(masks[None].contiguous(), plan, want_nviews=True)
    assert torch.equal(nv[0].cpu() > 0, torch.from_numpy(nviews[0] > 0))
    ic2, im2 = synthetic.images(cfg, cuda, seed=1)  # (iii)
    outs = m.evaluate_batch(torch.cat([ic, ic2]), torch.cat([im, im2]), [ids[0], ids[0]], [cams[0], cams[0]], [(S, S)] * 2, [(S, S)] * 2,
                            forced_new_tokens=forced)
    single2 = m.evaluate(ic2, im2, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)
    for o, one in zip(outs, (out, single2)):
        assert torch.equal(o["output_ids"], one["output_ids"])
        e = float((o["pred_contact_3d"] - one["pred_contact_3d"]).abs().max())
        assert e < 1e-3, e
    m.set_precision("parity")  # (ii)
    op = ev()
    par = op["pred_contact_3d"].float().cpu()
    _, nvp = ops.lift_mesh_plan(op["pred_masks"][0][None].contiguous(), plan, want_nviews=True)
    e = float((pc - par).abs().max())
    print(f"\n[13B model] default mode vs parity mode: max |dp| = {e:.2e}; resident LLaMA weights (default mode) {sum(rb.values()) / 1e9:.1f} GB")
    assert e < 1e-3 - 2e-5
    assert torch.equal(nv[0] > 0, nvp[0] > 0)
    for thr, cmp in ((0.5, torch.ge), (0.3, torch.gt)):
        same = cmp(pc, thr) == cmp(par, thr)
        band = (par - thr).abs() <= e + 2e-5
        assert bool(same[~band].all())
        print(f"[13B model] default: {int((~same).sum())} of {int(band.sum())} band vertices flip at {thr} (vs parity mode)")
