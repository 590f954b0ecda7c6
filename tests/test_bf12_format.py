"""The lossless 12-bit weight format of the decode linears: the numpy restatement (oracle/bf12.py) round-trips every bf16 bit pattern
class on the CPU, and (GPU) the library's packer writes byte for byte the planes the restatement describes."""
import numpy as np
import pytest


def _adversarial(N, K, seed):
    g = np.random.default_rng(seed)
    w = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bits = (w.view(np.uint32) >> 16).astype(np.uint16)  # (truncation: any bf16 pattern will do)
    bits[0, :8] = [0x0000, 0x8000, 0x0001, 0x807F, 0x2B8C, 0xB089, 0x3456, 0x0D22]  # zeros, subnormals, far below the window
    bits[1, 5] = 0x46EA  # 3e4: a row whose window excludes all its other weights
    bits[min(7, N - 1)] = 0
    bits[2, :] = 0x3F80  # a constant row
    return bits


@pytest.mark.parametrize("N,K,rows", [(32, 128, None), (35, 64, 48), (16, 1024, None)])
def test_numpy_restatement_is_lossless(N, K, rows):
    from oracle import bf12

    bits = _adversarial(N, K, N + K)
    p = bf12.pack(bits, rows)
    back = bf12.unpack(p)
    exp = np.where(bits == 0x8000, 0, bits)  # -0.0 packs as +0.0
    assert np.array_equal(back[:N], exp) and not back[N:].any()
    assert p["P"].size == p["shape"][0] * K and p["E"].size == p["shape"][0] * K // 2
    assert p["patch_ptr"][-1] == p["patch_col"].size and p["patch_ptr"][2] - p["patch_ptr"][1] == K - 1  # row 1: all but the 3e4
    x = np.random.default_rng(1).standard_normal(K)
    wf = (exp.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.allclose(bf12.gemv(p, x)[:N], wf @ x, rtol=1e-12, atol=1e-12)  # (float64 on identical weight values)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(48, 256), (35, 128), (4096, 4096)])
def test_library_packer_writes_the_restated_planes(hip_lib, cuda, N, K):
    import torch

    from interactvlm_amd import ops
    from oracle import bf12

    bits = _adversarial(N, K, N * 3 + K)
    w = torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).to(cuda)
    wp = ops.PackedBf12(w, pad_rows=True)
    ref = bf12.pack(bits, -(-N // 16) * 16)
    assert wp.frag and tuple(wp.shape) == ref["shape"] and wp.n_patches == ref["patch_col"].size
    assert np.array_equal(wp.P.flatten().cpu().numpy(), ref["P"]) and np.array_equal(wp.E.flatten().cpu().numpy(), ref["E"])
    assert np.array_equal(wp.ebase.cpu().numpy(), ref["ebase"]) and np.array_equal(wp.patch_ptr.cpu().numpy(), ref["patch_ptr"])
    n = wp.n_patches
    assert np.array_equal(wp.patch_col[:n].cpu().numpy(), ref["patch_col"])
    assert np.array_equal(wp.patch_val[:n].view(torch.int16).cpu().numpy().view(np.uint16), ref["patch_val"])
    x = torch.randn(1, K, generator=torch.Generator().manual_seed(3)).to(cuda)
    y = ops.linear_bf12(x, wp).double().cpu().numpy()[0]
    yr = bf12.gemv(ref, x.double().cpu().numpy()[0])[:N]
    assert np.abs(y - yr).max() <= 3e-6 * max(1.0, np.abs(yr).max())
