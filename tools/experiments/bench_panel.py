#!/usr/bin/env python3
"""SAM encoder GEMM shapes: row-major operands vs the K-panel layout (ivlm_gemm_bf16_panel), same kernels, bit-identical results.
    python tools/bench_panel.py [M]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import _lib, ops  # noqa: E402


def panelize(x):
    """[rows, K] -> K/64 panels of [rows, 64] (contiguous)"""
    r, k = x.shape
    return x.view(r, k // 64, 64).permute(1, 0, 2).contiguous()


def unpanel(p):
    kp, r, _ = p.shape
    return p.permute(1, 0, 2).reshape(r, kp * 64)


def t(f, n=20):
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    st = lambda: torch.cuda.current_stream().cuda_stream
    for name, N, K, act in (("qkv", 3840, 1280, "none"), ("proj", 1280, 1280, "none"), ("mlp1", 5120, 1280, "gelu"), ("mlp2", 1280, 5120, "none")):
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        ap, wp = panelize(a), panelize(w)
        out0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        outp = torch.empty(N // 64, M, 64, device=dev, dtype=torch.bfloat16)

        def row():
            ops.linear(a, w, b, act=act, out=out0)

        def pan(c_panel=False, a_panel=True):
            rc = lib.ivlm_gemm_bf16_panel(ap.data_ptr() if a_panel else a.data_ptr(), K, M * 64 if a_panel else 0, wp.data_ptr(), K, N * 64,
                                          outp.data_ptr() if c_panel else out1.data_ptr(), N, M * 64 if c_panel else 0, b.data_ptr(), None, 0,
                                          M, N, K, ops.ACT[act], 0, 0, None, None, st())
            assert rc == 0, rc

        row(); pan(); torch.cuda.synchronize()
        if os.environ.get("IVLM_LIB_PATH"):
            tp = t(pan)
            print(f"{name:5s} {os.path.basename(os.environ['IVLM_LIB_PATH'])}: {tp:7.1f} us", flush=True)
            continue
        assert torch.equal(out0, out1), float((out0.float() - out1.float()).abs().max())
        pan(True); torch.cuda.synchronize()
        assert torch.equal(unpanel(outp), out0)
        pan(False, False); torch.cuda.synchronize()
        assert torch.equal(out0, out1)

        tr, tp, tpc, tw = t(row), t(pan), t(lambda: pan(True)), t(lambda: pan(False, False))
        fl = 2.0 * M * N * K
        print(f"{name:5s} M={M} N={N} K={K}: row-major {tr:7.1f} us {fl / tr / 1e6:6.0f} TF | W panel only {tw:7.1f} us {fl / tw / 1e6:6.0f} TF | "
              f"A+W panel {tp:7.1f} us {fl / tp / 1e6:6.0f} TF | A+W+C panel {tpc:7.1f} us {fl / tpc / 1e6:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
