"""Attention kernel micro-benchmark on the path's shapes (+ fp32 torch check on a slice): SAM ViT-H global (4 views x 16 heads,
4096 tokens, d 80, rel-pos), SAM window (100 windows, 196 tokens), CLIP (257, d 64), LLaMA prefill (330, d 128, causal)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def ref_attn(q, k, v, scale, causal, rel, side):
    qf, kf, vf = q.float(), k.float(), v.float()
    if rel is not None:  # prescaled q rounded to bf16 first, like the kernel / SAM
        qf = (qf * scale).to(torch.bfloat16).float()
        s = qf @ kf.transpose(-1, -2)
        B, H, S, _ = q.shape
        rh, rw = rel
        bias = rh.view(B, H, S, side, 1) + rw.view(B, H, S, 1, side)
        s = s + bias.reshape(B, H, S, side * side)
    else:
        s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        Sq, Sk = s.shape[-2:]
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    return torch.softmax(s, -1) @ vf


def main():
    from interactvlm_amd import ops

    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    cases = [("sam_global", 4, 16, 4096, 80, False, 64), ("sam_window", 100, 16, 196, 80, False, 14),
             ("clip", 1, 16, 257, 64, False, 0), ("llama_prefill", 1, 32, 330, 128, True, 0)]
    for name, B, H, S, D, causal, side in cases:
        qkv = torch.randn(B, S, 3, H, D, generator=g, device=dev).to(bf)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        rel = None
        if side:
            rel = (0.5 * torch.randn(B * H, S, side, generator=g, device=dev), 0.5 * torch.randn(B * H, S, side, generator=g, device=dev))
        scale = D ** -0.5
        o = ops.attention(q, k, v, scale, causal=causal, rel=rel)
        nb = min(B, 2)
        r = ref_attn(q[:nb], k[:nb], v[:nb], scale, causal, None if rel is None else (rel[0][: nb * H], rel[1][: nb * H]), side)
        err = float((o[:nb].float() - r).abs().max())
        flops = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        row = [f"{name:14s} B={B:3d} H={H} S={S} D={D}: max|err| {err:.3e}"]
        from interactvlm_amd import _lib
        lib = _lib.load()
        for mode, label in ((0, "4-wave"), (1, "ping-pong")):
            lib.ivlm_attention_pingpong(mode)
            try:
                o2 = ops.attention(q, k, v, scale, causal=causal, rel=rel)
                e2 = float((o2[:nb].float() - r).abs().max())
                for _ in range(3):
                    ops.attention(q, k, v, scale, causal=causal, rel=rel)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                a.record()
                for _ in range(reps):
                    ops.attention(q, k, v, scale, causal=causal, rel=rel)
                b.record()
                torch.cuda.synchronize()
                us = a.elapsed_time(b) * 1e3 / reps
                row.append(f"{label}: {us:8.1f} us {flops / us * 1e-6:7.1f} TFLOP/s (err {e2:.1e})")
            finally:
                lib.ivlm_attention_pingpong(-1)
        print("  ".join(row), flush=True)

if __name__ == "__main__":
    main()
