#!/usr/bin/env python3
"""The hot MFMA GEMM shapes of one image as the DEFAULT precision mode issues them (fp16 operands): the five GEMMs of a SAM ViT-H
block at 4 views (M = 16384) and the four GEMMs of a LLaMA-7B prefill layer (M = 330, weights rotated so that they arrive cold from
HBM as in the pipeline).  One line per shape: microseconds (best of 3 rounds of 20 launches) and TFLOP/s.

Run once per library: the product build, and the ablation builds of tools/experiments/build_abl.sh (IVLM_LIB_PATH=tools/_bin/
libivlm_<tag>.so: epilogue / LDS fragment reads / DMA / MFMAs compiled out - their results are garbage by construction, only the time
counts).  tools/gemm_ceilings_table.py turns the logs into profiles/r05_gemm_ceilings.txt.

    python tools/bench_gemm_ceilings.py [--json out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


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
    from interactvlm_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    f16 = torch.float16
    rnd = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(f16).to(dev)
    rows = {}
    # ---- SAM ViT-H block, 4 views: 16384 tokens (windowed blocks carry 25 x 196-token windows per view: 19600 rows) ----
    M, D, MD = 16384, 1280, 5120
    xn2 = rnd(M, 2 * D)  # norm1 as [hi | lo] fp16 rows (the exact-q path)
    att, hh = rnd(M + 3216, D), rnd(M, MD)
    x = torch.randn(M, D, generator=g).to(dev)
    gather = torch.randperm(M + 3216, generator=g)[:M].to(torch.int32).to(dev)
    w = {"qkv": rnd(3 * D, D), "proj": rnd(D, D), "lin1": rnd(MD, D), "lin2": rnd(D, MD)}
    b = {k: (torch.randn(v.shape[0], generator=g) * 0.1).to(torch.bfloat16).to(dev) for k, v in w.items()}
    q2, kv = torch.empty(M, 2 * D, dtype=f16, device=dev), torch.empty(M, 2 * D, dtype=f16, device=dev)
    h_out = torch.empty(M, MD, dtype=f16, device=dev)
    sam = [
        ("sam q   16384x1280x1280 [hi|lo] A, [hi|lo] fp16 out", 2 * M * D * D,
         lambda: ops.linear(xn2, w["qkv"][:D], b["qkv"][:D], out=q2, a_split=True, out_split=True, out_f16=True)),
        ("sam k|v 16384x2560x1280 fp16 out", 2 * M * 2 * D * D, lambda: ops.linear(xn2[:, :D], w["qkv"][D:], b["qkv"][D:], out=kv)),
        ("sam proj 16384x1280x1280 row gather, fp32 residual in place", 2 * M * D * D,
         lambda: ops.linear(att, w["proj"], b["proj"], residual=x, out=x, a_rows=gather)),
        ("sam mlp1 16384x5120x1280 GELU fp16 out", 2 * M * MD * D, lambda: ops.linear(xn2[:, :D], w["lin1"], b["lin1"], act="gelu", out=h_out)),
        ("sam mlp2 16384x1280x5120 fp32 residual in place", 2 * M * MD * D, lambda: ops.linear(hh, w["lin2"], b["lin2"], residual=x, out=x)),
    ]
    for name, fl, fn in sam:
        us = timed(fn)
        rows[name] = {"us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}
        print(f"{name:64s} {us:8.1f} us {fl / us / 1e6:7.0f} TF", flush=True)
    del xn2, att, hh, x, w, q2, kv, h_out
    # ---- LLaMA-7B prefill layer, 330 positions, weights cold (4 rotating copies) ----
    M, H, I = 330, 4096, 11008
    a = rnd(M, H)
    hmid = rnd(M, I)
    xs = torch.randn(M, H, generator=g).to(dev)
    for name, N, K, act, res in (("llama q|k|v 330x12288x4096 fp16 out", 3 * H, H, "none", False),
                                 ("llama o     330x4096x4096 fp32 residual", H, H, "none", True),
                                 ("llama gate|up 330x22016x4096 SwiGLU fp16 out", 2 * I, H, "swiglu", False),
                                 ("llama down  330x4096x11008 fp32 residual", H, I, "none", True)):
        ws = [rnd(N, K) for _ in range(4)]
        it = [0]
        src = a if K == H else hmid

        def fn():
            wt = ws[it[0] % 4]
            it[0] += 1
            if res:
                ops.linear(src, wt, residual=xs, out_f32=True)
            else:
                ops.linear(src, wt, act=act, out_f16=True)
        us = timed(fn)
        fl = 2 * M * N * K
        rows[name] = {"us": round(us, 1), "tflops": round(fl / us / 1e6, 1), "splitk": ops._splitk_choice(M, N, K, act, None)}
        print(f"{name:64s} {us:8.1f} us {fl / us / 1e6:7.0f} TF  (split-K {rows[name]['splitk']})", flush=True)
        del ws
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({"lib": os.environ.get("IVLM_LIB_PATH", "product"), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
