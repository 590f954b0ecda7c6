#!/usr/bin/env python3
"""Weight-streaming rate of the decode GEMV on the LLaMA-7B shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import ops  # noqa: E402

SHAPES = [("qkv+rms", 12288, 4096, "none", True), ("o+res", 4096, 4096, "none", False),
          ("gateup+rms+swiglu", 22016, 4096, "swiglu", True), ("down+res", 4096, 11008, "none", False),
          ("lm_head", 32003, 4096, "none", False)]


def main():
    from interactvlm_amd import _lib
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    f32x = "--bf16x" not in sys.argv
    lib = _lib.load()
    for floor, ks in ((0, 0), (84 * 1024, 0), (0, 2), (0, 0), (84 * 1024, 0)):
      lib.ivlm_gemv1_lds_floor(floor)
      lib.ivlm_gemv1_tuning(ks)
      print(f"--- batch-1 kernel: LDS floor {floor}, ksplit {ks or 1}, x {'fp32' if f32x else 'bf16'}")
      for name, N, K, act, rms in SHAPES:
        run_shape(dev, bf, f32x, name, N, K, act, rms)


def run_shape(dev, bf, f32x, name, N, K, act, rms):
    if True:
        # rotate over several weight copies so the 256 MB Infinity Cache cannot serve the stream
        ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(bf) for _ in range(ncopy)]
        x = torch.randn(1, K, device=dev)
        if not f32x:
            x = x.to(bf)
        g = torch.ones(K, device=dev).to(bf)
        res = torch.randn(1, N, device=dev) if act == "none" and not rms and N == 4096 else None
        if res is not None and not f32x:
            res = res.to(bf)
        out_f32 = f32x or name == "lm_head"
        kw = dict(act=act, residual=res, rms=(g, 1e-5) if rms else None, out_f32=out_f32)
        for w in ws:
            ops.linear(x, w, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        s.record()
        for _ in range(iters):
            for w in ws:
                ops.linear(x, w, **kw)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / (iters * ncopy) * 1e-3
        print(f"{name:20s} N={N:6d} K={K:6d} {t*1e6:8.1f} us  {N*K*2/t/1e12:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
