#!/usr/bin/env python3
"""[r6] LLaMA prefill GEMMs (M = 330), fp16 operands, K-panel weights, cold weights (4 rotating copies): the product tiling against the
352-row tiles of round 6 (tile codes 352 = 352 x 128, 353 = 352 x 256; one block of 8 waves per CU, W fetched once per CU).
    python tools/experiments/bench_prefill_tiles_r6.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import _lib, ops  # noqa: E402
from interactvlm_amd.ops import ACT, _p, _stream, check  # noqa: E402


def t(f, n=24):
    for _ in range(4):
        f()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            f()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 330
    F16, PANEL, RESF32 = ops.GEMM_F16, ops.GEMM_W_PANEL, ops.GEMM_RES_F32
    for name, N, K, act, f32 in (("qkv", 12288, 4096, "none", False), ("o", 4096, 4096, "none", True),
                                 ("gateup", 22016, 4096, "swiglu", False), ("down", 4096, 11008, "none", True)):
        ws = [ops.panel_weight((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.float16)) for _ in range(4)]
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.float16)
        n_out = N // 2 if act == "swiglu" else N
        res = torch.randn(M, n_out, device=dev) if f32 else None
        out = torch.empty(M, n_out, device=dev, dtype=torch.float32 if f32 else torch.float16)
        wsp = torch.empty(16 * M * N, dtype=torch.float32, device=dev)
        it = [0]
        ref = {}

        def run(tile, splits):
            w = ws[it[0] % 4]
            it[0] += 1
            lib.ivlm_gemm_tile_override(tile)
            flags = F16 | PANEL | (RESF32 if f32 else 0) | (0 if f32 else ops.GEMM_OUT_F16)
            ldw = 64  # (ignored with IVLM_GEMM_W_PANEL)
            if splits == 1:
                check(lib.ivlm_gemm_bf16(a.data_ptr(), K, w.data_ptr(), ldw, out.data_ptr(), n_out, None, _p(res), n_out, 0, M, N, K, ACT[act],
                                         1 if f32 else 0, 1, 0, 0, 0, 0, None, 0.0, flags, None, None, _stream()), "gemm")
            else:
                check(lib.ivlm_gemm_bf16_splitk(a.data_ptr(), K, w.data_ptr(), ldw, out.data_ptr(), n_out, None, _p(res), n_out, 0, M, N, K,
                                                ACT[act], 1 if f32 else 0, splits, wsp.data_ptr(), wsp.numel() * 4, flags, _stream()), "splitk")

        def same(tile, sp):  # every tiling must give the product's numbers (same weights: copy 0)
            it[0] = 0
            run(tile, sp)
            torch.cuda.synchronize()
            o = out.float().clone()
            if "o" not in ref:
                ref["o"] = o
                return 0.0
            return float((o - ref["o"]).abs().max())

        auto = ops._splitk_choice(M, N, K, act, None)
        same(0, auto)
        line = f"{name:7s} N={N} K={K}: product (splits {auto}) {t(lambda: run(0, auto)):6.1f} us |"
        for tile in (176, 352, 353):
            for sp in ((1,) if act == "swiglu" else (1, 2, 4, 8, 16)):
                if K % (sp * 64) or (K // sp) < 256:
                    continue
                d = same(tile, sp)
                line += f" t{tile}/s{sp} {t(lambda: run(tile, sp)):6.1f}" + (f" (!diff {d:.1e})" if d > 2e-2 else "")
        lib.ivlm_gemm_tile_override(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
