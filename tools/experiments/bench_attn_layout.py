#!/usr/bin/env python3
"""SAM attention (window 196 tokens x 100 windows; global 4096 tokens x 4 views; 16 heads of 80) on q/k/v read from the fused
[token][3][head][80] buffer the q|k|v GEMM writes today vs a head-major [3*head][token][80] buffer (contiguous 160-byte rows per head)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import ops  # noqa: E402


def t(f, n=20):
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
    dev = torch.device("cuda:0")
    H, D = 16, 80
    for name, B, S, side in (("window", 100, 196, 14), ("global", 4, 4096, 64)):
        g = torch.Generator(device=dev).manual_seed(0)
        tok = torch.randn(B, S, 3, H, D, generator=g, device=dev).to(torch.bfloat16)      # today's layout
        hm = tok.permute(2, 3, 0, 1, 4).contiguous()                                        # [3, H, B, S, D]
        qa, ka, va = (tok[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        qb, kb, vb = (hm[i].permute(1, 0, 2, 3) for i in range(3))                          # [B, H, S, D] views
        rel = (0.5 * torch.randn(B * H, S, side, generator=g, device=dev), 0.5 * torch.randn(B * H, S, side, generator=g, device=dev))
        scale = D ** -0.5
        oa = ops.attention(qa, ka, va, scale, rel=rel)
        ob = ops.attention(qb, kb, vb, scale, rel=rel)
        assert torch.equal(oa, ob)
        ta = t(lambda: ops.attention(qa, ka, va, scale, rel=rel))
        tb = t(lambda: ops.attention(qb, kb, vb, scale, rel=rel))
        blk = {"h": (0.1 * torch.randn(2 * side - 1, D, generator=g, device=dev)).to(torch.bfloat16), }
        tabw = (0.1 * torch.randn(2 * side - 1, D, generator=g, device=dev)).to(torch.bfloat16)
        cat = ops.relpos_tables_cat(blk["h"], tabw)
        ra = t(lambda: ops.relpos_bias(qa, blk["h"], tabw, side, side, cat=cat))
        rb = t(lambda: ops.relpos_bias(qb, blk["h"], tabw, side, side, cat=cat))
        print(f"{name}: attention {ta:7.1f} us (token-major) vs {tb:7.1f} us (head-major); rel-pos rows {ra:6.1f} vs {rb:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
