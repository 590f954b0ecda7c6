#!/usr/bin/env python3
"""TFLOP/s of the fused attention kernel on the hot-path shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import ops  # noqa: E402

SHAPES = [("sam_global", 4, 16, 4096, 4096, 80, False, True), ("sam_window", 100, 16, 196, 196, 80, False, True),
          ("sam_global_norel", 4, 16, 4096, 4096, 80, False, False),
          ("clip", 1, 16, 257, 257, 64, False, False), ("llm_prefill", 1, 32, 330, 330, 128, True, False),
          ("llm_decode", 1, 32, 1, 354, 128, True, False), ("dec_t2i", 4, 8, 9, 4096, 16, False, False),
          ("dec_i2t", 4, 8, 4096, 9, 16, False, False)]


def main():
    dev = torch.device("cuda:0")
    for name, B, H, Sq, Sk, D, causal, rel in SHAPES:
        q = torch.randn(B, H, Sq, D, device=dev).to(torch.bfloat16)
        k = torch.randn(B, H, Sk, D, device=dev).to(torch.bfloat16)
        v = torch.randn(B, H, Sk, D, device=dev).to(torch.bfloat16)
        r = None
        if rel:
            side = int(Sk ** 0.5)
            r = (torch.randn(B * H, Sq, side, device=dev), torch.randn(B * H, Sq, side, device=dev))
        out = ops.attention(q, k, v, D ** -0.5, causal=causal, q_pos0=Sk - Sq, rel=r)
        for _ in range(3):
            ops.attention(q, k, v, D ** -0.5, causal=causal, q_pos0=Sk - Sq, rel=r, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        s.record()
        for _ in range(iters):
            ops.attention(q, k, v, D ** -0.5, causal=causal, q_pos0=Sk - Sq, rel=r, out=out)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / iters * 1e-3
        fl = 4.0 * B * H * Sq * Sk * D * (0.5 if causal and Sq > 1 else 1.0)
        print(name, f"{t*1e6:.1f} us", f"{fl/t/1e12:.1f} TFLOP/s", flush=True)


def bench_relpos():
    dev = torch.device("cuda:0")
    for name, B, H, SH in (("relpos_global", 4, 16, 64), ("relpos_window", 100, 16, 14)):
        q = torch.randn(B, H, SH * SH, 80, device=dev).to(torch.bfloat16)
        th = torch.randn(2 * SH - 1, 80, device=dev).to(torch.bfloat16)
        ops.relpos_bias(q, th, th, SH, SH)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.relpos_bias(q, th, th, SH, SH)
        e.record()
        torch.cuda.synchronize()
        print(name, f"{s.elapsed_time(e) / 10 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    bench_relpos()
    main()
