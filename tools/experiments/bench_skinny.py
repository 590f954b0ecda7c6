"""Skinny-M linears of the batched decode step: wave-per-row GEMV (M <= 8) / tile GEMM (M > 8) vs the split-K MFMA kernel
(csrc/gemv_mfma.hip), LLaMA-7B shapes, M = 1..16.  us per launch and weight-streaming GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    shapes = [("qkv", 12288, 4096, "none", True), ("o", 4096, 4096, "none", False), ("gu", 22016, 4096, "swiglu", True),
              ("down", 4096, 11008, "none", False), ("lm_head", 32000, 4096, "none", False)]
    nrot = 6  # rotate weight copies so the matrix is not L2 / MALL resident between launches
    for name, N, K, act, rms in shapes:
        ws = [torch.randn(N, K, device=dev).to(bf) / K ** 0.5 for _ in range(nrot)]
        gam = torch.ones(K, device=dev, dtype=bf)
        for M in (1, 2, 3, 4, 6, 8, 12, 16):
            x = torch.randn(M, K, device=dev).to(bf)
            row = [f"{name:8s} M={M:2d}"]
            for mode, min_m in (("default-old", 17), ("mfma", 1)):
                if mode == "mfma" and False:
                    continue
                lib.ivlm_gemv_mfma_min_m(min_m)
                kw = dict(act=act, rms=(gam, 1e-5) if (rms and (M <= 8 or min_m == 1)) else None)
                try:
                    for i in range(3):
                        ops.linear(x, ws[i % nrot], **kw)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 30
                    a.record()
                    for i in range(reps):
                        ops.linear(x, ws[i % nrot], **kw)
                    b.record()
                    torch.cuda.synchronize()
                    us = a.elapsed_time(b) * 1e3 / reps
                    row.append(f"{mode}: {us:7.1f} us {N * K * 2 / us * 1e-3:6.0f} GB/s")
                except Exception as e:  # noqa: BLE001
                    row.append(f"{mode}: {type(e).__name__}")
                finally:
                    lib.ivlm_gemv_mfma_min_m(0)
            print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
