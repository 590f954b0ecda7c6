#!/usr/bin/env python3
"""LLaMA prefill GEMMs (M = 330 rows): block tile x split-K sweep.  python tools/bench_prefill_gemm.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import _lib, ops  # noqa: E402
from interactvlm_amd.ops import ACT, _p, _stream, check  # noqa: E402


def t(f, n=30):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 330
    for name, N, K, act, f32 in (("qkv", 12288, 4096, "none", False), ("o", 4096, 4096, "none", True), ("gateup", 22016, 4096, "swiglu", False),
                                 ("down", 4096, 11008, "none", True)):
        # rotate weights so that they come from HBM as in the pipeline
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(4)]
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        n_out = N // 2 if act == "swiglu" else N
        res = torch.randn(M, n_out, device=dev) if f32 else None
        out = torch.empty(M, n_out, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        wsp = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
        it = [0]

        def run(tile, splits):
            w = ws[it[0] % 4]
            it[0] += 1
            lib.ivlm_gemm_tile_override(tile)
            flags = 2 if f32 else 0
            if splits == 1:
                check(lib.ivlm_gemm_bf16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), n_out, None, _p(res), n_out, 0, M, N, K, ACT[act],
                                         1 if f32 else 0, 1, 0, 0, 0, 0, None, 0.0, flags, None, None, _stream()), "gemm")
            else:
                check(lib.ivlm_gemm_bf16_splitk(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), n_out, None, _p(res), n_out, 0, M, N, K,
                                                ACT[act], 1 if f32 else 0, splits, wsp.data_ptr(), wsp.numel() * 4, flags, _stream()), "splitk")
        line = f"{name:7s} N={N} K={K}:"
        auto = ops._splitk_choice(M, N, K, act, None)
        line += f" auto(tile 0, splits {auto}) {t(lambda: run(0, auto)):6.1f} us |"
        for tile in (64, 128, 176, 352):
            for sp in ((1,) if act == "swiglu" else (1, 2, 4, 8)):
                if K % (sp * 64):
                    continue
                line += f" t{tile}/s{sp} {t(lambda: run(tile, sp)):6.1f}"
        lib.ivlm_gemm_tile_override(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
