"""Batch-1 decode linears of LLaMA-7B: the bf16 GEMV against the lossless 12-bit packed layout (ivlm_gemv1_bf12), back to back per
shape (warm caches do not matter: every matrix is far larger than the Infinity Cache only for gate|up; cold numbers come from the
decode loop itself - tools/bench_decode.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)

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

    flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for name, N, K, act, rms, res in (("q|k|v", 12288, 4096, "none", True, False), ("o_proj", 4096, 4096, "none", False, True),
                                      ("gate|up", 22016, 4096, "swiglu", True, False), ("down", 4096, 11008, "none", False, True),
                                      ("lm_head", 32003, 4096, "none", False, False)):
        w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).bfloat16()
        wp = ops.PackedBf12(w)
        x = torch.randn(1, K, generator=g, device=dev)
        gam = torch.ones(K, dtype=torch.bfloat16, device=dev)
        r = torch.zeros(1, N, device=dev) if res else None
        kw = dict(act=act, residual=r, rms=(gam, 1e-5) if rms else None)
        t16 = t(lambda: ops.linear(x, w, out_f32=True, **kw))
        t12 = t(lambda: ops.linear_bf12(x, wp, **kw))

        def cold(f):  # with a 1-GB write between launches: weights come from HBM, as in the decode loop
            tz = t(lambda: flush.zero_(), 5)
            return t(lambda: (flush.zero_(), f()), 5) - tz
        c16, c12 = cold(lambda: ops.linear(x, w, out_f32=True, **kw)), cold(lambda: ops.linear_bf12(x, wp, **kw))
        mb = N * K * 2 / 1e6
        print(f"{name:8s} {N}x{K}: bf16 {t16:6.1f} us ({mb / t16 * 1e-3:.2f} TB/s)  bf12 {t12:6.1f} us ({mb / t12 * 1e-3:.2f} TB/s of bf16 bytes, "
              f"{wp.bytes() / 1e6 / t12 * 1e-3:.2f} moved)  cold: {c16:6.1f} -> {c12:6.1f} us  patches {wp.n_patches}", flush=True)


if __name__ == "__main__":
    main()
