"""GPU parity of the dense building blocks (bf16 MFMA GEMM + epilogues, LayerNorm, RMSNorm) against
plain PyTorch fp32 on the same bf16-rounded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf(t):
    import torch

    return t.to(torch.bfloat16)


def _ref_act(x, act):
    import torch
    import torch.nn.functional as F

    if act == "gelu":
        return F.gelu(x)
    if act == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if act == "relu":
        return F.relu(x)
    if act == "silu":
        return F.silu(x)
    return x


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 1024, 1024), (330, 4096, 4096), (1000, 1280, 5120),
                                   (9, 256, 256), (1, 128, 2048), (4096, 128, 256), (131, 36, 192), (300, 480, 160), (77, 64, 8)])
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_gemm_shapes(hip_lib, cuda, M, N, K, act):
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = _bf(torch.randn(N, generator=g) * 0.1)
    r = _bf(torch.randn(M, N, generator=g))
    ref = _ref_act(x.float() @ w.float().T + b.float(), act) + r.float()
    got = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda))
    assert got.dtype == torch.bfloat16 and got.shape == (M, N)
    err = (got.float().cpu() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 4e-3  # one bf16 ulp of the result + accumulation-order noise
    assert bool((err <= tol).all()), f"max err {err.max().item()} at {M}x{N}x{K}"
    got32 = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act=act, residual=r.to(cuda), out_f32=True)
    assert got32.dtype == torch.float32
    assert torch.allclose(got32.cpu(), ref, atol=3e-3, rtol=1e-3)


@pytest.mark.parametrize("M,N,K", [(330, 4096, 11008), (257, 1024, 4096), (257, 1024, 1024), (64, 256, 2048)])
def test_gemm_splitk_matches_single_pass(hip_lib, cuda, M, N, K):
    """Small-M shapes take the split-K path (ops._splitk_choice > 1): same numbers as the one-pass kernel up to fp32
    summation order, and both within a bf16 ulp of the fp32 reference."""
    import torch

    from interactvlm_amd import ops

    assert ops._splitk_choice(M, N, K, "none", None) > 1
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    b = _bf(torch.randn(N, generator=g) * 0.1).to(cuda)
    r = _bf(torch.randn(M, N, generator=g)).to(cuda)
    for act in ("none", "quick_gelu"):
        split = ops.linear(x, w, b, act=act, residual=r, out_f32=True)
        ops.SPLITK = False
        try:
            single = ops.linear(x, w, b, act=act, residual=r, out_f32=True)
        finally:
            ops.SPLITK = True
        ref = _ref_act(x.float() @ w.float().T + b.float(), act) + r.float()
        assert torch.allclose(split, single, atol=2e-5, rtol=1e-5)
        assert torch.allclose(split, ref, atol=2e-4, rtol=1e-4)
    out = torch.full((M, N + 8), 7.0, dtype=torch.bfloat16, device=cuda)  # strided output rows (ldc > N)
    ops.linear(x, w, out=out[:, :N])
    assert bool((out[:, N:] == 7.0).all())
    assert torch.allclose(out[:, :N].float(), x.float() @ w.float().T, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("tile", [64, 96, 128, 256, 512, 320])
def test_gemm_forced_tiles(hip_lib, cuda, tile):
    """Both block-tile configurations on a shape with ragged M/N edges and a K tail."""
    import torch

    from interactvlm_amd import _lib, ops

    g = torch.Generator().manual_seed(tile)
    M, N, K = 700, 900, 1096
    x = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = _bf(torch.randn(N, generator=g) * 0.1)
    r = _bf(torch.randn(M, N, generator=g))
    ref = _ref_act(x.float() @ w.float().T + b.float(), "gelu") + r.float()
    prev = _lib.load().ivlm_gemm_tile_override(tile)
    try:
        got = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act="gelu", residual=r.to(cuda), out_f32=True)
    finally:
        _lib.load().ivlm_gemm_tile_override(prev)
    assert torch.allclose(got.cpu(), ref, atol=3e-3, rtol=1e-3)


@pytest.mark.parametrize("tile", [512, 320])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 128), (4096, 1280, 1280), (1000, 520, 200), (2048, 3840, 1288)])
def test_gemm_8phase_matches_simple_kernel_bitwise(hip_lib, cuda, M, N, K, tile):
    """The 8-phase ping-pong 256^2 kernel (tile code 512) and the 256 x 320 kernel (320, gemm320.hip) accumulate in the same
    order as the plain double-buffered 256^2 kernel, so they must agree with it BIT FOR BIT; repeated launches screen for LDS races in the staggered pipeline
    (a late DMA / early read shows up as a few wrong tiles in some runs)."""
    import torch

    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    b = _bf(torch.randn(N, generator=g) * 0.1).to(cuda)
    prev = lib.ivlm_gemm_tile_override(256)
    try:
        base = ops.linear(x, w, b, act="gelu", out_f32=True)
        base_bf = ops.linear(x, w, b, act="none")
        lib.ivlm_gemm_tile_override(tile)  # eight waves, 8-phase loop (gemm256.hip)
        assert torch.equal(ops.linear(x, w, b, act="none"), base_bf)  # the bf16 store path of the same kernels
        filler = torch.randn(64 << 20, device=cuda)  # perturb memory timing between runs
        for it in range(30):
            got = ops.linear(x, w, b, act="gelu", out_f32=True)
            if it % 3 == 0:
                filler.mul_(1.0001)
            assert torch.equal(got, base), f"run {it}: {(got != base).sum().item()} elements differ"
    finally:
        lib.ivlm_gemm_tile_override(prev)
    ref = _ref_act(x.float() @ w.float().T + b.float(), "gelu")
    assert torch.allclose(base, ref, atol=3e-3, rtol=1e-3)


@pytest.mark.parametrize("N,K,act,rms,res,f32", [(12288, 4096, "none", True, False, False), (22016, 4096, "swiglu", True, False, False),
                                                 (4096, 11008, "none", False, True, False), (4096, 4096, "none", False, True, False),
                                                 (32003, 4096, "none", False, False, True), (1000, 1376, "quick_gelu", True, True, False),
                                                 (5120, 512, "none", False, False, False)])
def test_gemv_decode_shapes_vs_fp32(hip_lib, cuda, N, K, act, rms, res, f32):
    """M == 1 decode GEMVs (bf16 activations) against fp32 torch; K = 11008 / 1376 rows are not a whole number of waves
    long (two rows per wave step); repeated launches are bit-equal."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(N + K)
    x = _bf(torch.randn(1, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    gam = _bf(1 + 0.1 * torch.randn(K, generator=g)).to(cuda)
    n_out = N // 2 if act == "swiglu" else N
    r = _bf(torch.randn(1, n_out, generator=g)).to(cuda) if res else None
    kw = dict(act=act, residual=r, rms=(gam, 1e-5) if rms else None, out_f32=f32)
    got = ops.linear(x, w, **kw)
    assert torch.equal(got, ops.linear(x, w, **kw))
    xf = x.float()
    if rms:  # the fused path: bf16(x * gamma) . w, scaled by rstd afterwards
        xf = (xf * gam.float()).to(torch.bfloat16).float() * torch.rsqrt(x.float().pow(2).mean() + 1e-5)
    y = xf @ w.float().T
    if act == "swiglu":
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    else:
        y = _ref_act(y, act)
    if res:
        y = y + r.float()
    tol = dict(atol=2e-3, rtol=1e-3) if f32 else dict(atol=2e-2, rtol=1.6e-2)
    assert torch.allclose(got.float(), y, **tol)


@pytest.mark.parametrize("M", [1, 2, 4, 8, 13, 16])
@pytest.mark.parametrize("N,K,act,rms,res,bias", [
    (12288, 4096, "none", True, False, False),   # q|k|v with the fused RMSNorm
    (22016, 4096, "swiglu", True, False, False),  # gate|up
    (4096, 11008, "none", False, True, False),    # down + fp32 residual (K * 4 bytes = 44 KB of LDS at M = 1)
    (256, 4096, "relu", False, False, True),      # text_hidden_fcs-like: N < 1024 keeps M <= 8 on the GEMV
    (264, 8, "sigmoid", False, False, True),      # cam-pose first layer (K padded to 8)
])
def test_fp32_activation_linear_is_exact_in_products(hip_lib, cuda, M, N, K, act, rms, res, bias):
    """fp32 activation rows on the weight-streaming kernels (decode path, cam encoders, text_hidden_fcs): bf16-weight x
    fp32-activation products are exact in the GEMV (M == 1, or small matrices) and exact to 2^-17 on the skinny MFMA kernel
    (hi + lo operand split) - the result matches an fp64 reference to fp32 accumulation noise, NOT to bf16 operand noise."""
    import torch

    from interactvlm_amd import ops

    if M > 8 and (N < 1024 or K < 1024):
        pytest.skip("M > 8 needs the skinny MFMA kernel (K, N >= 1024)")
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    gam = _bf(1 + 0.1 * torch.randn(K, generator=g))
    b = _bf(0.1 * torch.randn(N, generator=g)) if bias else None
    n_out = N // 2 if act == "swiglu" else N
    r = torch.randn(M, n_out, generator=g) if res else None
    got = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda) if bias else None, act=act, residual=r.to(cuda) if res else None,
                     rms=(gam.to(cuda), 1e-5) if rms else None, out_f32=True)
    assert got.dtype == torch.float32 and got.shape == (M, n_out)
    xd = x.double()
    if rms:
        xd = xd * gam.double() * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5)
    y = xd @ w.double().T
    if bias:
        y = y + b.double()
    if act == "swiglu":
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    else:
        y = _ref_act(y, act)
    if res:
        y = y + r.double()
    err = (got.double().cpu() - y).abs().max().item()
    scale = y.abs().max().item()
    assert err < 3e-5 * max(scale, 1.0), (err, scale)  # bf16 operand rounding would give ~4e-3 * scale


def test_gemm_fp32_residual_and_scatter_epilogue(hip_lib, cuda):
    """fp32 residual stream (IVLM_GEMM_RES_F32) on the tile GEMM, split-K and the 8-phase kernel; the scatter epilogue
    (out_rows: SAM window_unpartition + shortcut folded into the proj GEMM), in place on the residual buffer."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(11)
    for M, N, K in ((300, 512, 256), (330, 4096, 4096), (4096, 1280, 1280)):
        x = _bf(torch.randn(M, K, generator=g)).to(cuda)
        w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
        b = _bf(0.1 * torch.randn(N, generator=g)).to(cuda)
        r = torch.randn(M, N, generator=g).to(cuda)
        got = ops.linear(x, w, b, residual=r, out_f32=True)
        ref = x.float() @ w.float().T + b.float() + r
        assert torch.allclose(got, ref, atol=2e-3, rtol=1e-4), (M, N, K)
    # scatter: 6 "windows" of 5 rows, some rows are padding (-1), the valid ones a permutation of the 24 stream rows
    M, N, K, R = 30, 64, 128, 24
    perm = torch.randperm(R, generator=g)
    rows = torch.full((M,), -1, dtype=torch.int32)
    rows[torch.randperm(M, generator=g)[:R]] = perm.to(torch.int32)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    stream = torch.randn(R, N, generator=g).to(cuda)
    exp = stream.clone()
    y = x.float() @ w.float().T
    for m in range(M):
        if rows[m] >= 0:
            exp[rows[m]] += y[m]
    out = ops.linear(x, w, residual=stream, out=stream, out_rows=rows.to(cuda))
    assert out.data_ptr() == stream.data_ptr() and torch.allclose(stream, exp, atol=1e-3, rtol=1e-4)
    # the same scatter through the 256 x 320 tile (whole-line part and the directly stored fifth fragment column)
    M3, N3, K3, R3 = 700, 640, 192, 600
    rows3 = torch.full((M3,), -1, dtype=torch.int32)
    rows3[torch.randperm(M3, generator=g)[:R3]] = torch.randperm(R3, generator=g).to(torch.int32)
    x3 = _bf(torch.randn(M3, K3, generator=g)).to(cuda)
    w3 = _bf(torch.randn(N3, K3, generator=g) / K3 ** 0.5).to(cuda)
    stream3 = torch.randn(R3, N3, generator=g).to(cuda)
    exp3 = stream3.clone()
    y3 = x3.float() @ w3.float().T
    valid = rows3 >= 0
    exp3[rows3[valid].long().to(cuda)] += y3[valid.to(cuda)]
    from interactvlm_amd import _lib as _l
    prev = _l.load().ivlm_gemm_tile_override(320)
    try:
        ops.linear(x3, w3, residual=stream3, out=stream3, out_rows=rows3.to(cuda))
    finally:
        _l.load().ivlm_gemm_tile_override(prev)
    assert torch.allclose(stream3, exp3, atol=1e-3, rtol=1e-4)
    # gather prologue: product row r reads A row a_rows[r] (proj of a windowed SAM block on the real rows only), every tile
    for tile, (M2, N2, K2) in ((0, (24, 64, 128)), (128, (300, 256, 192)), (512, (700, 512, 256)), (64, (500, 192, 64)),
                                (320, (700, 640, 256)), (0, (4096, 1280, 192))):
        src = _bf(torch.randn(M2 + 37, K2, generator=g)).to(cuda)
        amap = torch.randperm(M2 + 37, generator=g)[:M2].to(torch.int32)
        w2 = _bf(torch.randn(N2, K2, generator=g) / K2 ** 0.5).to(cuda)
        res = torch.randn(M2, N2, generator=g).to(cuda)
        from interactvlm_amd import _lib
        prev = _lib.load().ivlm_gemm_tile_override(tile)
        try:
            got = ops.linear(src, w2, residual=res, out_f32=True, a_rows=amap.to(cuda))
        finally:
            _lib.load().ivlm_gemm_tile_override(prev)
        ref = src.float()[amap.long().to(cuda)] @ w2.float().T + res
        assert got.shape == (M2, N2) and torch.allclose(got, ref, atol=2e-3, rtol=1e-4), (tile, M2, N2, K2)


@pytest.mark.parametrize("M,N,K,act,f32res", [(16384, 1280, 1280, "none", True), (16384, 1280, 5120, "none", True),
                                              (8192, 768, 512, "gelu", False), (4096, 1536, 256, "none", False)])
def test_gemm_column_split_matches_unsplit(hip_lib, cuda, M, N, K, act, f32res):
    """The column split of under-filled 256 x 256 rounds (SAM proj / mlp2: 320 tiles -> 256 on the 8-phase kernel + a strip
    on 128 x 64 tiles) against the unsplit launch: same K order per output element on every tile kernel, so the two agree to
    fp32 rounding of different MFMA groupings; both against fp32 torch."""
    import torch

    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    b = _bf(0.1 * torch.randn(N, generator=g)).to(cuda)
    r = torch.randn(M, N, generator=g).to(cuda) if f32res else _bf(torch.randn(M, N, generator=g)).to(cuda)
    kw = dict(act=act, residual=r, out_f32=True)
    got = ops.linear(x, w, b, **kw)
    lib.ivlm_gemm_nsplit(0)
    try:
        one = ops.linear(x, w, b, **kw)
    finally:
        lib.ivlm_gemm_nsplit(1)
    ref = _ref_act(x.float() @ w.float().T + b.float(), act) + r.float()
    assert torch.allclose(got, ref, atol=3e-3, rtol=1e-4) and torch.allclose(one, ref, atol=3e-3, rtol=1e-4)
    assert torch.allclose(got, one, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("cols", [256, 1280, 4096])
def test_norms_fp32_in_out_and_row_map(hip_lib, cuda, cols):
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(cols)
    x = torch.randn(37, cols, generator=g) * 3 + 0.5
    w = _bf(1 + 0.1 * torch.randn(cols, generator=g))
    b = _bf(0.1 * torch.randn(cols, generator=g))
    ref = F.layer_norm(x, (cols,), w.float(), b.float(), 1e-6)
    got = ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), 1e-6, out_f32=True).cpu()
    assert got.dtype == torch.float32 and torch.allclose(got, ref, atol=2e-5, rtol=1e-5)
    got_b = ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), 1e-6).cpu()
    assert got_b.dtype == torch.bfloat16
    assert float((got_b.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    ref_r = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()  # fp32 stream: no intermediate bf16 cast
    got_r = ops.rmsnorm(x.to(cuda), w.to(cuda), 1e-5, out_f32=True).cpu()
    assert torch.allclose(got_r, ref_r, atol=2e-5, rtol=1e-5)
    # row map: row r -> out_rows[r] of a larger, pre-zeroed buffer (window_partition folded into norm1)
    perm = torch.randperm(50, generator=g)[:37].to(torch.int32)
    buf = torch.zeros(50, cols, dtype=torch.bfloat16, device=cuda)
    ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), 1e-6, out=buf, out_rows=perm.to(cuda))
    exp = torch.zeros(50, cols, dtype=torch.bfloat16)
    exp[perm.long()] = got_b
    assert torch.equal(buf.cpu(), exp)


def test_row_kernels_dtypes_and_split(hip_lib, cuda):
    """gather_rows / add_rows across bf16 / fp32 / split outputs; split rows: hi + lo == x to 2^-17 and a GEMM against
    [W | W] reproduces the fp32-activation product."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(2)
    src = torch.randn(20, 64, generator=g)
    idx = torch.tensor([3, -1, 7, 19, 0, -1, 5], dtype=torch.int32)
    add = torch.randn(7, 64, generator=g)
    exp = torch.where(idx[:, None] >= 0, src[idx.clamp(min=0).long()], torch.zeros(1)) + add
    got = ops.gather_rows(src.to(cuda), idx.to(cuda), add=add.to(cuda))
    assert got.dtype == torch.float32 and torch.equal(got.cpu(), exp)
    got_b = ops.gather_rows(_bf(src).to(cuda), idx.to(cuda), add=add.to(cuda))  # bf16 src + fp32 add -> fp32
    assert got_b.dtype == torch.float32 and torch.equal(
        got_b.cpu(), torch.where(idx[:, None] >= 0, _bf(src).float()[idx.clamp(min=0).long()], torch.zeros(1)) + add)
    conv = ops.gather_rows(src.to(cuda), out_kind="bf16")  # identity + conversion
    assert torch.equal(conv.cpu(), _bf(src))
    sp = ops.split_rows(src.to(cuda)).cpu()
    assert sp.shape == (20, 128) and torch.equal(sp[:, :64], _bf(src))
    assert float((sp[:, :64].float() + sp[:, 64:].float() - src).abs().max()) <= 2.0 ** -16 * float(src.abs().max())
    a = torch.randn(12, 64, generator=g)
    tab = torch.randn(5, 64, generator=g)
    for op, f in (("add", lambda u, v: u + v), ("mul", lambda u, v: u * v)):
        e = f(a, tab[torch.arange(12) % 5])
        assert torch.equal(ops.add_rows(a.to(cuda), tab.to(cuda), op=op).cpu(), e)
        assert torch.equal(ops.add_rows(a.to(cuda), tab.to(cuda), op=op, out_kind="bf16").cpu(), _bf(e))
        s2 = ops.add_rows(a.to(cuda), tab.to(cuda), op=op, out_kind="split").cpu()
        assert torch.equal(s2[:, :64], _bf(e)) and float((s2[:, :64].float() + s2[:, 64:].float() - e).abs().max()) < 1e-4
    # fp32-activation GEMM through the split: error ~2^-17 relative instead of ~2^-9
    x = torch.randn(300, 256, generator=g)
    w = _bf(torch.randn(128, 256, generator=g) / 16)
    w2 = torch.cat([w, w], 1).contiguous()
    got = ops.linear(ops.split_rows(x.to(cuda)), w2.to(cuda), out_f32=True).cpu()
    ref = (x.double() @ w.double().T).float()
    assert float((got - ref).abs().max()) < 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize("B,H,Sq,Sk,D,div", [(4, 8, 9, 4096, 16, 1), (4, 8, 4096, 9, 16, 1), (4, 8, 9, 9, 32, 1),
                                            (4, 8, 9, 9, 32, 4), (2, 1, 4, 4, 128, 1), (3, 2, 70, 130, 32, 1)])
def test_attention_f32_vs_torch(hip_lib, cuda, B, H, Sq, Sk, D, div):
    """ops.attention_f32 (SAM mask decoder / AttentionSplitter attentions, fp32 operands) against torch fp64."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(B + Sq + Sk + D)
    q = torch.randn(B, Sq, H, D, generator=g).to(cuda).permute(0, 2, 1, 3)
    k = torch.randn(B // div, Sk, H, D, generator=g).to(cuda).permute(0, 2, 1, 3)
    v = torch.randn(B // div, Sk, H, D, generator=g).to(cuda).permute(0, 2, 1, 3)
    got = ops.attention_f32(q, k, v, D ** -0.5)
    kk = k.repeat_interleave(div, 0).double()
    vv = v.repeat_interleave(div, 0).double()
    ref = torch.softmax(q.double() @ kk.transpose(-1, -2) * D ** -0.5, -1) @ vv
    assert got.shape == (B, H, Sq, D) and got.transpose(1, 2).is_contiguous()
    assert float((got.double() - ref).abs().max()) < 2e-5


def test_argmax_first_index_ties_and_unaligned_rows(hip_lib, cuda):
    """torch.argmax semantics (first index of the maximum) on rows that do not start on 16-byte boundaries."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(3)
    for cols in (1, 3, 7, 255, 4099, 32003):
        x = torch.randn(5, cols, generator=g)
        x[1, cols // 2:] = x[1].max() + 1.0   # a run of equal maxima: the FIRST must win
        x[2, -1] = 100.0                      # maximum in the scalar tail
        x[3, 0] = 100.0                       # maximum in the scalar head
        got = ops.argmax(x.to(cuda)).cpu()
        assert got.tolist() == x.argmax(-1).tolist(), cols


def test_gemm_transpose_detecting(hip_lib, cuda):
    """A = I (padded) with an ASYMMETRIC W catches row/col swaps in the MFMA C layout."""
    import torch

    from interactvlm_amd import ops

    K = 128
    x = torch.zeros(96, K)
    x[torch.arange(96), torch.arange(96)] = 1.0
    w = (torch.arange(200)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.01)
    got = ops.linear(_bf(x).to(cuda), _bf(w).to(cuda), out_f32=True).cpu()
    assert torch.allclose(got, _bf(w).float().T[:96], atol=1e-6)


@pytest.mark.parametrize("act", ["quick_gelu", "relu", "silu"])
def test_gemm_activations_no_bias(hip_lib, cuda, act):
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(300, 256, generator=g))
    w = _bf(torch.randn(512, 256, generator=g) / 16)
    ref = _ref_act(x.float() @ w.float().T, act)
    got = ops.linear(x.to(cuda), w.to(cuda), act=act, out_f32=True).cpu()
    assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3)


def test_gemm_swiglu_and_rowmod_residual(hip_lib, cuda):
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(5)
    M, K, I = 77, 512, 384
    x = _bf(torch.randn(M, K, generator=g))
    wg = _bf(torch.randn(I, K, generator=g) / K ** 0.5)
    wu = _bf(torch.randn(I, K, generator=g) / K ** 0.5)
    inter = torch.stack([wg, wu], dim=1).reshape(2 * I, K).contiguous()  # rows (gate_j, up_j)
    ref = F.silu(x.float() @ wg.float().T) * (x.float() @ wu.float().T)
    got = ops.linear(x.to(cuda), inter.to(cuda), act="swiglu", out_f32=True).cpu()
    assert got.shape == (M, I)
    assert torch.allclose(got, ref, atol=3e-3, rtol=2e-3)
    # residual broadcast with row modulo (pos_embed-style table)
    tab = _bf(torch.randn(11, 2 * I, generator=g))
    ref2 = x.float() @ inter.float().T + tab.float()[torch.arange(M) % 11]
    got2 = ops.linear(x.to(cuda), inter.to(cuda), residual=tab.to(cuda), res_mod=11, out_f32=True).cpu()
    assert torch.allclose(got2, ref2, atol=3e-3, rtol=2e-3)


def test_gemm_strided_input_rows(hip_lib, cuda):
    """Row stride > K (reading q out of a fused qkv buffer)."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(9)
    big = _bf(torch.randn(50, 3 * 128, generator=g)).to(cuda)
    w = _bf(torch.randn(64, 128, generator=g) / 11).to(cuda)
    x = big[:, 128:256]
    got = ops.linear(x, w, out_f32=True).cpu()
    assert torch.allclose(got, x.float().cpu() @ w.float().cpu().T, atol=3e-3, rtol=2e-3)


@pytest.mark.parametrize("cols,eps", [(256, 1e-5), (1024, 1e-5), (1280, 1e-6), (4096, 1e-5), (5120, 1e-5)])
def test_layernorm_rmsnorm(hip_lib, cuda, cols, eps):
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(cols)
    x = _bf(torch.randn(37, cols, generator=g) * 3 + 0.5)
    w = _bf(1 + 0.1 * torch.randn(cols, generator=g))
    b = _bf(0.1 * torch.randn(cols, generator=g))
    ref = F.layer_norm(x.float(), (cols,), w.float(), b.float(), eps)
    got = ops.layernorm(x.to(cuda), w.to(cuda), b.to(cuda), eps).float().cpu()
    assert torch.allclose(got, ref, atol=2e-2, rtol=2 ** -7)
    xf = x.float()
    n = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16)
    ref_r = (w * n).float()  # HF LlamaRMSNorm: bf16 weight * bf16 normalised
    got_r = ops.rmsnorm(x.to(cuda), w.to(cuda), eps).float().cpu()
    assert torch.allclose(got_r, ref_r, atol=2e-2, rtol=2 ** -7)


@pytest.mark.parametrize("M", [3, 5, 8, 13, 16])
@pytest.mark.parametrize("N,K,act,rms,res,f32,bias", [
    (4096, 4096, "none", False, True, False, False),    # o_proj
    (2752, 1024, "swiglu", True, False, False, False),  # gate|up with the fused RMSNorm
    (1032, 1096, "gelu", False, False, True, True),     # N not a multiple of 16, K not a multiple of 32 x 8 waves
    (1024, 11008, "none", True, True, False, False),    # down-projection length (43 k-steps per wave)
])
def test_skinny_mfma_gemm_vs_fp32(hip_lib, cuda, M, N, K, act, rms, res, f32, bias):
    """gemv_mfma.hip (batched decode: up to 16 activation rows, split-K inside the block, MFMA) against fp32 torch and,
    where it applies (M <= 8), the wave-per-row GEMV; deterministic across launches."""
    import torch

    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    gam = _bf(1 + 0.1 * torch.randn(K, generator=g)).to(cuda)
    b = _bf(0.1 * torch.randn(N, generator=g)).to(cuda) if bias else None
    n_out = N // 2 if act == "swiglu" else N
    r = _bf(torch.randn(M, n_out, generator=g)).to(cuda) if res else None
    kw = dict(act=act, residual=r, rms=(gam, 1e-5) if rms else None, out_f32=f32, bias=b)
    lib.ivlm_gemv_mfma_min_m(1)
    try:
        got = ops.linear(x, w, **kw)
        assert torch.equal(got, ops.linear(x, w, **kw))
    finally:
        lib.ivlm_gemv_mfma_min_m(0)
    xf = x.float()
    if rms:
        xf = (xf * gam.float()).to(torch.bfloat16).float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5)
    y = xf @ w.float().T
    if bias:
        y = y + b.float()
    if act == "swiglu":
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    else:
        y = _ref_act(y, act)
    if res:
        y = y + r.float()
    tol = dict(atol=2e-3, rtol=1e-3) if f32 else dict(atol=2e-2, rtol=1.6e-2)
    assert got.shape == (M, n_out) and torch.allclose(got.float(), y, **tol)
    if M <= 8 and not (rms and M * K * 2 > 48 * 1024):  # (RMS-fused rows that do not fit LDS exist on the skinny kernel only)
        lib.ivlm_gemv_mfma_min_m(17)  # never: the wave-per-row kernel
        try:
            roww = ops.linear(x, w, **kw)
        finally:
            lib.ivlm_gemv_mfma_min_m(0)
        assert torch.allclose(got.float(), roww.float(), **tol)


@pytest.mark.parametrize("M,N,K,act", [(4096, 3840, 1280, "none"), (2048, 5120, 1280, "gelu"), (4096, 1280, 5120, "none"),
                                       (330, 4096, 512, "none"), (1000, 576, 192, "gelu")])
def test_gemm_k_panel_layouts_equal_row_major(hip_lib, cuda, M, N, K, act):
    """ivlm_gemm_bf16_panel: operands and / or output stored as K/64 panels of [rows][64] (1 KB contiguous per wave DMA
    instruction) - same kernels, same arithmetic: BIT-identical to the row-major call, for every combination of panelised A, W
    and C, on the 8-phase 256^2 kernel, the column split (K = 5120) and the small-tile kernel (M = 330 - a shape the row-major call does not split over K -, ragged 1000 x 576)."""
    import torch

    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g) * 0.5).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    b = _bf(torch.randn(N, generator=g) * 0.1).to(cuda)
    pan = lambda x: x.view(x.shape[0], x.shape[1] // 64, 64).permute(1, 0, 2).contiguous()
    unpan = lambda p: p.permute(1, 0, 2).reshape(p.shape[1], p.shape[0] * 64)
    ap, wp = pan(a), pan(w)
    ref = ops.linear(a, w, b, act=act)
    st = torch.cuda.current_stream().cuda_stream
    for a_p, w_p, c_p in ((1, 1, 0), (0, 1, 0), (1, 0, 0), (1, 1, 1), (0, 0, 1)):
        out = torch.empty(N // 64, M, 64, device=cuda, dtype=torch.bfloat16) if c_p else torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
        rc = lib.ivlm_gemm_bf16_panel((ap if a_p else a).data_ptr(), K, M * 64 if a_p else 0, (wp if w_p else w).data_ptr(), K,
                                      N * 64 if w_p else 0, out.data_ptr(), N, M * 64 if c_p else 0, b.data_ptr(), None, 0, M, N, K,
                                      ops.ACT[act], 0, 0, None, None, st)
        assert rc == 0, (a_p, w_p, c_p, rc)
        got = unpan(out) if c_p else out
        assert torch.equal(got, ref), (a_p, w_p, c_p, float((got.float() - ref.float()).abs().max()))
    # argument checks: K must be whole panels, panel strides must cover the rows, fp32 output has no panel form
    out = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    assert lib.ivlm_gemm_bf16_panel(ap.data_ptr(), K, M * 64 - 8, wp.data_ptr(), K, N * 64, out.data_ptr(), N, 0, None, None, 0, M, N, K,
                                    0, 0, 0, None, None, st) == -1
    assert lib.ivlm_gemm_bf16_panel(ap.data_ptr(), K, M * 64, wp.data_ptr(), K, N * 64, out.data_ptr(), N, M * 64, None, None, 0, M, N, K,
                                    0, 1, 0, None, None, st) == -4


@pytest.mark.parametrize("M,N,K,act,rms", [(8, 12288, 4096, "none", True), (8, 22016, 4096, "swiglu", True), (5, 4096, 11008, "none", False),
                                           (16, 32003, 1024, "none", False)])
def test_skinny_tiles_per_block_do_not_change_the_result(hip_lib, cuda, M, N, K, act, rms):
    """gemv_mfma.hip: a block may own 1, 2, 3, 4 or 6 tiles of 16 weight rows (the automatic choice is 3 for the fused q|k|v rows
    and 6 for gate|up); every tile is accumulated by the same waves in the same order, so the result is bit-identical."""
    import torch

    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    gam = _bf(1 + 0.1 * torch.randn(K, generator=g)).to(cuda)
    kw = dict(act=act, rms=(gam, 1e-5) if rms else None, out_f32=True)
    lib.ivlm_gemv_mfma_min_m(1)
    try:
        base = ops.linear(x, w, **kw)  # automatic
        for t in (1, 2, 3, 4, 6):
            lib.ivlm_skinny_tuning(t)
            assert torch.equal(ops.linear(x, w, **kw), base), t
    finally:
        lib.ivlm_skinny_tuning(0)
        lib.ivlm_gemv_mfma_min_m(0)


@pytest.mark.parametrize("M,N,K", [(12, 256, 256), (9, 256, 4096), (16, 2048, 256), (12, 256, 8), (13, 128, 128)])
def test_fp32_rows_9_to_16_small_matrices(hip_lib, cuda, M, N, K):
    """9..16 fp32 activation rows against matrices the skinny MFMA kernel does not take (K or N < 1024: cam-pose
    encoders, AttentionSplitter with n_seg * V > 8, text_hidden_fcs[1] with 9-16 [SEG] rows, narrow decode batches) run as row
    chunks on the weight-streaming GEMV - exact bf16-weight x fp32-activation products, all epilogues."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M * 31 + N + K)
    x = torch.randn(M, K, generator=g)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = _bf(torch.randn(N, generator=g) * 0.1)
    r = torch.randn(M, N, generator=g)
    ref = torch.relu(x.double() @ w.double().T + b.double()) + r.double()
    got = ops.linear(x.to(cuda), w.to(cuda), b.to(cuda), act="relu", residual=r.to(cuda), out_f32=True)
    assert got.shape == (M, N) and float((got.cpu().double() - ref).abs().max()) < 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,act", [(330, 12288, 4096, "none"), (330, 22016, 4096, "swiglu"), (200, 8192, 1024, "none"),
                                       (352, 8192, 512, "swiglu"), (129, 8320, 4096, "none")])
def test_row_stationary_prefill_tile(hip_lib, cuda, M, N, K, act):
    """The 176 x 128 row-stationary tile of the LLaMA prefill (128 < M <= 352, N >= 8192; q|k|v with two K slices): same numbers
    as the 128 x 64 tiling and within a bf16 ulp of fp32."""
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import _lib, ops

    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(cuda)
    w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    ref = x.float() @ w.float().T
    if act == "swiglu":
        ref = F.silu(ref[:, 0::2]) * ref[:, 1::2]
    got = ops.linear(x, w, act=act, out_f32=True)
    lib = _lib.load()
    lib.ivlm_gemm_tile_override(64)
    ops.SPLITK = False
    try:
        base = ops.linear(x, w, act=act, out_f32=True)
    finally:
        ops.SPLITK = True
        lib.ivlm_gemm_tile_override(0)
    assert torch.allclose(got, base, atol=3e-5, rtol=1e-5)
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4)
    gb = ops.linear(x, w, act=act)
    assert gb.dtype == torch.bfloat16 and torch.allclose(gb.float(), ref, atol=2e-2, rtol=1e-2)


def test_gemm_f16_split_operand_and_output(hip_lib, cuda):
    """fp16-mode "exact q" projection: A = [hi | lo] IEEE halves (layernorm(out_split, out_f16)), fp16 weight, the fp32 result written
    as [hi | lo] IEEE halves - against fp64 on the fp32 activations; and the hi half alone is the plain fp16 row."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(5)
    M, K, N = 1000, 1280, 1280
    x = torch.randn(M, K, generator=g).to(cuda)
    lw, lb = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(cuda), (0.1 * torch.randn(K, generator=g)).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(cuda)
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().to(cuda)
    xs = ops.layernorm(x, lw, lb, 1e-6, out_split=True, out_f16=True)
    x1 = ops.layernorm(x, lw, lb, 1e-6, out_f16=True)
    assert xs.dtype == torch.float16 and xs.shape == (M, 2 * K) and torch.equal(xs[:, :K], x1)
    ref_n = torch.nn.functional.layer_norm(x.double(), (K,), lw.double(), lb.double(), 1e-6)
    assert (xs[:, :K].double() + xs[:, K:].double() - ref_n).abs().max().item() < 2e-6
    w16 = ops.f16_weight(w)
    y = ops.linear(xs, w16, b, a_split=True, out_split=True, out_f16=True)
    assert y.dtype == torch.float16 and y.shape == (M, 2 * N)
    ref = ref_n @ w.double().t() + b.double()
    err = (y[:, :N].double() + y[:, N:].double() - ref).abs().max().item()
    y1 = ops.linear(x1, w16, b, out_f16=True)
    err1 = (y1.double() - ref).abs().max().item()
    print(f"fp16 split GEMM: max err {err:.2e} (single fp16 operand / output: {err1:.2e})")
    assert err < 2e-5 and err1 > 10 * err
    # scatter epilogue + strided hi-half operand (the k|v GEMM of the exact-q path reads the hi half in place)
    y2 = ops.linear(xs[:, :K], w16, b, out_f16=True)
    assert torch.equal(y2, y1)


@pytest.mark.parametrize("N,K,act,rms,res", [(12288, 4096, "none", True, False), (4096, 4096, "none", False, True),
                                             (22016, 4096, "swiglu", True, False), (4096, 11008, "none", False, True),
                                             (32003, 4096, "none", False, False), (40, 512, "none", True, True),
                                             (48, 1024, "swiglu", True, False), (16, 64, "none", False, True),
                                             (5120, 13824, "none", False, True)])
def test_gemv_bf12_is_lossless_and_equals_the_bf16_gemv(hip_lib, cuda, N, K, act, rms, res):
    """The 12-bit packed weight layout of the batch-1 decode linears: every weight (zeros, subnormals, the far tail below the row's
    exponent window: patches) is reconstructed BIT FOR BIT, and the linear on the packed matrix equals the bf16 GEMV on the original
    one up to fp32 summation order (both: exact bf16 x fp32 products) - checked against fp64."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    # adversarial entries: exact zeros, negative zero, bf16 subnormals, values 20 - 40 binades under the row maximum, one huge row
    w[0, :8] = torch.tensor([0.0, -0.0, 1e-40, -3e-39, 1e-12, -1e-9, 2e-7, 1e-30]).bfloat16()
    w[1, 5] = 3e4
    w[min(7, N - 1)] = 0
    w = w.to(cuda)
    x = (torch.randn(1, K, generator=g) * 2.0).to(cuda)
    gam = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(cuda) if rms else None
    r = torch.randn(1, N, generator=g).to(cuda) if res else None
    kw = dict(act=act, residual=r, rms=(gam, 1e-5) if rms else None)
    ref16 = ops.linear(x, w, out_f32=True, **kw)
    got = None
    if N % 16 == 0 and K % 64 == 0:  # the library's packer (what a C caller uses) == the torch restatement of the format, byte for byte
        a_, b_ = ops.PackedBf12(w, packer="c"), ops.PackedBf12(w, packer="torch")
        assert a_.frag and b_.frag and a_.n_patches == b_.n_patches
        for f in ("P", "E", "ebase", "patch_ptr"):
            assert torch.equal(getattr(a_, f).flatten(), getattr(b_, f).flatten()), f
        assert torch.equal(a_.patch_col[: a_.n_patches], b_.patch_col[: a_.n_patches])
        assert torch.equal(a_.patch_val[: a_.n_patches].view(torch.int16), b_.patch_val[: a_.n_patches].view(torch.int16))
    for fragments in (False, True):  # row layout (VALU kernel) and fragment layout (MFMA kernel: N % 16 == 0, K % 64 == 0)
        wp = ops.PackedBf12(w, fragments=fragments)
        assert wp.frag == (fragments and N % 16 == 0 and K % 64 == 0)
        assert torch.equal(wp.unpack().view(torch.int16), torch.where(w == 0, torch.zeros_like(w), w).view(torch.int16))  # (-0.0 -> +0.0)
        assert wp.n_patches >= 6 and (N < 1000 or wp.bytes() < 0.77 * w.numel() * 2)  # (1.5 of 2 bytes per weight + tables / patches)
        y_ = ops.linear_bf12(x, wp, **kw)
        if got is not None:
            assert float((y_ - got).abs().max()) <= 3e-6 * max(1.0, float(got.abs().max()))
        got = y_
    if N % 16 == 0 and K % 64 == 0:  # the batched decode step: M <= 16 activation rows on the same planes (hi + lo bf16 operands: 2^-17)
        wpf = ops.PackedBf12(w)
        for M in (2, 7, 16):
            xm = (torch.randn(M, K, generator=g) * 2.0).to(cuda)
            rm = torch.randn(M, N, generator=g).to(cuda) if res else None
            kwm = dict(act=act, residual=rm, rms=(gam, 1e-5) if rms else None)
            ym = ops.linear_bf12(xm, wpf, **kwm)
            xdm = xm.double()
            if rms:
                xdm = xdm * torch.rsqrt((xdm * xdm).mean(dim=1, keepdim=True) + 1e-5) * gam.double()
            rf = xdm @ w.double().t()
            if act == "swiglu":
                rf = torch.nn.functional.silu(rf[:, 0::2]) * rf[:, 1::2]
            if res:
                rf = rf + rm.double()
            em = float((ym.double() - rf).abs().max()) / float(rf.abs().max())
            assert ym.shape == rf.shape and em < 3e-5, (M, em)
            y16 = ops.linear(xm, w, out_f32=True, **kwm)  # the bf16-weight skinny kernel: same operands, other summation order
            assert float((ym - y16).abs().max()) / float(rf.abs().max()) < 3e-5
    if N % 16 and K % 64 == 0 and not res:  # rows padded to 16 with zeros (the lm_head): fragment layout, N outputs
        wpad = ops.PackedBf12(w, pad_rows=True)
        assert wpad.frag and wpad.rows == N and wpad.shape[0] % 16 == 0
        assert torch.equal(wpad.unpack().view(torch.int16), torch.where(w == 0, torch.zeros_like(w), w).view(torch.int16))
        y_ = ops.linear_bf12(x, wpad, **kw)
        assert y_.shape == got.shape and float((y_ - got).abs().max()) <= 3e-6 * max(1.0, float(got.abs().max()))
    xd = x.double()
    if rms:
        xd = xd * torch.rsqrt((xd * xd).mean() + 1e-5) * gam.double()
    y = xd @ w.double().t()
    if act == "swiglu":
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    if res:
        y = y + r.double()
    scale = float(y.abs().max())
    e12, e16 = float((got.double() - y).abs().max()) / scale, float((ref16.double() - y).abs().max()) / scale
    print(f"\\n[bf12 GEMV {N}x{K} {act}] vs fp64: packed {e12:.2e}, bf16 GEMV {e16:.2e}; patches {wp.n_patches}")
    assert got.shape == ref16.shape and e12 < 3e-6 and e16 < 3e-6
    assert float((got - ref16).abs().max()) / scale < 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize("big", [1.0e9, 3.0e15, 1.0e-12])
def test_gemv_bf12_activation_range(hip_lib, cuda, big):
    """the packed kernels stage x times an exact power of two.  With 2^100 one activation of 1e9 overflowed fp32 and the
    whole output row turned NaN, a range the bf16-weight GEMV (plain fp32 x) does not have.  The scale is now 2^64 - the middle of
    fp32's exponent range: rows with one element around 1e9 / 3e15, and a row of tiny activations (1e-12), come out finite and equal
    to fp64 like the bf16 GEMV's - batch-1 MFMA form, VALU form and the M <= 16 form."""
    import torch

    from interactvlm_amd import ops

    N, K = 256, 1024
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(cuda)
    x = torch.randn(1, K, generator=g)
    if big >= 1.0:
        x[0, 17] = big
    else:
        x = x * big
    x = x.to(cuda)
    ref = x.double() @ w.double().t()
    scale = float(ref.abs().max())
    y16 = ops.linear(x, w, out_f32=True)
    for fragments in (True, False):
        y = ops.linear_bf12(x, ops.PackedBf12(w, fragments=fragments))
        assert bool(torch.isfinite(y).all()), fragments
        assert float((y.double() - ref).abs().max()) / scale < 3e-6, fragments
        assert float((y - y16).abs().max()) / scale < 3e-6
    xm = x.repeat(3, 1) * torch.tensor([[1.0], [-0.5], [2.0]], device=cuda)
    ym = ops.linear_bf12(xm, ops.PackedBf12(w))
    assert bool(torch.isfinite(ym).all())
    assert float((ym.double() - xm.double() @ w.double().t()).abs().max()) / (2 * scale) < 3e-5


@pytest.mark.parametrize("M,N,K,res", [(330, 4096, 4096, True), (257, 1024, 4096, True), (330, 12288, 4096, False), (40, 256, 1024, False)])
def test_fused_splitk_equals_two_launch_splitk(hip_lib, cuda, M, N, K, res):
    """ivlm_gemm_bf16_splitk_fused (the tile's last-arriving block sums the K slices inside the GEMM launch; arrival counters that
    every launch leaves at zero) == ivlm_gemm_bf16_splitk (a reduction launch) BIT FOR BIT - the same slices summed in the same order
    - on fp16 and bf16 operands, with fp32-residual / 16-bit outputs, repeatedly on the same counters."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M + N)
    for dt in (torch.float16, torch.bfloat16):
        x = torch.randn(M, K, generator=g).to(dt).to(cuda)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(cuda)
        r = torch.randn(M, N, generator=g).to(cuda) if res else None
        kw = dict(residual=r, out_f32=True) if res else (dict(out_f16=True) if dt == torch.float16 else {})
        if ops._splitk_choice(M, N, K, "none", None) < 2:
            pytest.skip("shape not split")
        prev = ops.SPLITK_FUSED
        try:
            ops.SPLITK_FUSED = False
            ref = ops.linear(x, w, **kw)
            ops.SPLITK_FUSED = True
            for _ in range(3):
                got = ops.linear(x, w, **kw)
                assert torch.equal(got, ref)
            assert int(ops._splitk_counters(x.device).abs().sum()) == 0  # left at zero
        finally:
            ops.SPLITK_FUSED = prev
    ref64 = x.double() @ w.double().t() + (r.double() if res else 0)
    assert float((got.double() - ref64).abs().max()) / float(ref64.abs().max()) < 1e-2


@pytest.mark.parametrize("M,N,K,act,res", [(330, 12288, 4096, "none", False), (330, 4096, 4096, "none", True), (330, 22016, 4096, "swiglu", False),
                                           (330, 4096, 11008, "none", True), (5280, 4096, 4096, "none", True), (40, 512, 256, "none", False)])
def test_k_panel_weights_equal_row_major(hip_lib, cuda, M, N, K, act, res):
    """IVLM_GEMM_W_PANEL: the weight stored as K/64 panels of [N][64] (ops.panel_weight) through every tiling the
    LLaMA prefill uses - 176 x 128 with and without K slices, 128 x 64 with four K slices, the 256-row tiles of the packed 16-prompt
    prefill - gives the row-major result BIT FOR BIT (same tiles, same arithmetic, other addresses), fp16 and bf16, SwiGLU epilogue
    and fp32 residual."""
    import torch

    from interactvlm_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    for dt in (torch.float16, torch.bfloat16):
        x = torch.randn(M, K, generator=g).to(dt).to(cuda)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(cuda)
        r = torch.randn(M, N, generator=g).to(cuda) if res else None
        kw = dict(residual=r, out_f32=True) if res else dict(act=act)
        wp = ops.panel_weight(w)
        assert wp.shape == (K // 64, N, 64) and torch.equal(wp.permute(1, 0, 2).reshape(N, K), w)
        assert torch.equal(ops.linear(x, wp, **kw), ops.linear(x, w, **kw)), dt
    with pytest.raises(ops.IvlmError):
        ops.linear(x[:8], wp)  # (M <= 16: the weight-streaming kernels take row-major weights)
