#!/usr/bin/env python3
"""[r6] The SAM encoder's two LayerNorms at 4 views (16384 x 1280 fp32 rows -> [hi | lo] fp16 / fp16) and the LLaMA / CLIP norms:
the 32-bytes-per-lane kernel against the line-contiguous one (ivlm_norm_line_loads).  python tools/experiments/bench_norm_r6.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import _lib, ops  # noqa: E402


def t(f, n=40):
    for _ in range(5):
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
    for name, rows, cols, kind, out_bytes in (("sam norm1 f32 -> [hi|lo] f16", 16384, 1280, "f16_split", 4), ("sam norm2 f32 -> f16", 16384, 1280, "f16", 2),
                                              ("sam 64 views norm2", 16384 * 16, 1280, "f16", 2), ("clip ln f32 -> f16", 257, 1024, "f16", 2)):
        xs = [torch.randn(rows, cols, device=dev) for _ in range(3)]  # rotate: 84 MB each, beyond what stays in L2
        w = torch.ones(cols, device=dev, dtype=torch.bfloat16)
        b = torch.zeros(cols, device=dev, dtype=torch.bfloat16)
        it = [0]
        outs = {}

        def run():
            it[0] += 1
            return ops.layernorm(xs[it[0] % 3], w, b, 1e-6, **({"out_split": True, "out_f16": True} if kind == "f16_split" else {"out_f16": True}))
        line = f"{name:32s} {rows} x {cols}:"
        for on in (0, 1, 0, 1):
            lib.ivlm_norm_line_loads(on)
            it[0] = 0
            o = run()
            outs[on] = o.float().clone()
            us = t(run)
            mb = rows * cols * (4 + out_bytes) / 1e6
            line += f"  line_loads={on}: {us:6.1f} us ({mb / us * 1e-3:.2f} TB/s)"
        d = float((outs[0] - outs[1]).abs().max())
        print(line + f"  max |diff| {d:.1e}", flush=True)
    lib.ivlm_norm_line_loads(1)


if __name__ == "__main__":
    main()
