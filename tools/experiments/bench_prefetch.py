#!/usr/bin/env python3
"""The prefetch wave of the small-M tile GEMMs (gemm_bf16_kernel<.., PFW>): LLaMA-7B prefill shapes (M = 330, cold weights: 4 rotating
copies) and CLIP ViT-L shapes (M = 257) for prefetch distances 0 (off) .. 6.    python tools/bench_prefetch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, n=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(n):
            fn()
        e_.record()
        torch.cuda.synchronize()
        best = min(best, s_.elapsed_time(e_) / n * 1e3)
    return best


def main():
    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    f16 = torch.float16
    rnd = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(f16).to(dev)
    dists = [0, 1, 2, 3, 4, 6]
    for name, M, N, K, act, res in (("llama q|k|v", 330, 12288, 4096, "none", False), ("llama o", 330, 4096, 4096, "none", True),
                                    ("llama gate|up", 330, 22016, 4096, "swiglu", False), ("llama down", 330, 4096, 11008, "none", True),
                                    ("clip qkv", 257, 3072, 1024, "none", False), ("clip fc2", 257, 1024, 4096, "none", True)):
        ws = [rnd(N, K) for _ in range(4)]
        a = rnd(M, K)
        xs = torch.randn(M, N, generator=g).to(dev) if res else None
        it = [0]

        def fn():
            wt = ws[it[0] % 4]
            it[0] += 1
            if res:
                ops.linear(a, wt, residual=xs, out_f32=True)
            else:
                ops.linear(a, wt, act=act, out_f16=True)
        line = f"{name:14s} {M}x{N}x{K}:"
        ref = None
        for d in dists:
            lib.ivlm_gemm_prefetch(d)
            out = ops.linear(a, ws[0], residual=xs, out_f32=True) if res else ops.linear(a, ws[0], act=act, out_f16=True)
            if ref is None:
                ref = out
            assert torch.equal(out, ref), (name, d)  # the prefetch wave changes nothing but the time
            line += f"  d{d} {timed(fn):6.1f}"
        lib.ivlm_gemm_prefetch(3)
        print(line + " us", flush=True)
        del ws


if __name__ == "__main__":
    main()
