#!/usr/bin/env python3
"""TFLOP/s of the bf16 MFMA GEMM on the hot-path shapes (HIP events, random data)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import ops  # noqa: E402

SHAPES = [  # (name, M, N, K)
    ("sam_qkv", 16384, 3840, 1280), ("sam_proj", 16384, 1280, 1280), ("sam_mlp1", 16384, 5120, 1280),
    ("sam_mlp2", 16384, 1280, 5120), ("sam_qkv_win", 19600, 3840, 1280),
    ("llm_qkv", 330, 12288, 4096), ("llm_o", 330, 4096, 4096), ("llm_gateup", 330, 22016, 4096),
    ("llm_down", 330, 4096, 11008), ("clip_qkv", 257, 3072, 1024), ("clip_fc1", 257, 4096, 1024),
    ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
]


def main():
    from interactvlm_amd import _lib
    dev = torch.device("cuda:0")
    res = {}
    tile = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
    _lib.load().ivlm_gemm_tile_override(tile)
    _lib.load().ivlm_gemm_nsplit(0 if "--nosplit" in sys.argv else 1)
    print("tile override:", tile, "nsplit:", "--nosplit" not in sys.argv)
    if "--fp8" in sys.argv:  # the same shapes with e4m3 operands (K elements = K bytes)
        one = torch.ones(1, device=dev)
        for name, M, N, K in SHAPES:
            if M <= 16:
                continue
            xq = torch.randint(0, 100, (M, K), dtype=torch.uint8, device=dev)
            wq = torch.randint(0, 100, (N, K), dtype=torch.uint8, device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                ops.linear_fp8(xq, wq, one, one, out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.linear_fp8(xq, wq, one, one, out=out)
            e.record()
            torch.cuda.synchronize()
            t = s.elapsed_time(e) / 20 * 1e-3
            print(name, "fp8", {"us": round(t * 1e6, 1), "TFLOPs": round(2 * M * N * K / t / 1e12, 1)}, flush=True)
        return
    for name, M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.linear(x, w, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        s.record()
        for _ in range(iters):
            ops.linear(x, w, out=out)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / iters * 1e-3
        res[name] = {"us": round(t * 1e6, 1), "TFLOPs": round(2 * M * N * K / t / 1e12, 1),
                     "GBps": round((M * K + N * K + M * N) * 2 / t / 1e9, 1)}
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
