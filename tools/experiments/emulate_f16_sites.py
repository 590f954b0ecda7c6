#!/usr/bin/env python3
"""Diagnostic (plain torch on the GPU box; never the product path): which operand-rounding SITES of a SAM ViT block move the
encoder output, and by how much, when an operand is rounded to IEEE fp16 (or bf16).  The oracle's fp32 encoder arithmetic
(oracle/nn.py sam_block, restated here with rounding hooks) runs in fp64 on one view of the headline ViT-H with seeded weights,
once exactly and once per site set; prints the relative rms error of the [256, 64, 64] embedding.

    sites: n1q / n1kv (norm1 out as seen by the q / the k|v columns of the q|k|v GEMM), qrel / qqk (the q GEMM output as seen by the rel-pos terms / by Q.K^T), k v (GEMM outputs), qs (q * scale rounded again), rel (rel-pos terms),
           p (softmax weights), o (attention output -> proj input), n2 (norm2 out), h (GELU hidden)
    python tools/emulate_f16_sites.py [f16|bf16] [depth]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SITES = ("n1q", "n1kv", "qrel", "qqk", "k", "v", "qs", "rel", "p", "o", "n2", "h")


def main():
    from interactvlm_amd import synthetic
    from interactvlm_amd.weights import SAM_PREFIX

    kind = sys.argv[1] if len(sys.argv) > 1 else "f16"
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    c = cfg.sam
    wall = synthetic.device_weights(cfg, dev, seed=3)
    p = SAM_PREFIX + ".image_encoder"
    w = {k: v.double() for k, v in wall.items() if k.startswith(p)}
    del wall
    _, im = synthetic.images(cfg, dev, seed=5)
    img = im[0, :1].double()
    rt = torch.float16 if kind == "f16" else torch.bfloat16
    H = c.num_heads

    def run(active):
        r = lambda name, t: t.to(rt).double() if name in active else t
        lin = lambda pre, x: x @ w[pre + ".weight"].t() + w[pre + ".bias"]
        ln = lambda pre, x: F.layer_norm(x, (x.shape[-1],), w[pre + ".weight"], w[pre + ".bias"], 1e-6)

        def rel_tab(size, tab):
            i = torch.arange(size, device=dev)
            return tab[(i[:, None] - i[None, :]) + size - 1]

        def attention(pre, x):
            B, Hh, Wd, C = x.shape
            hd = C // H
            qkv = lin(pre + ".qkv", r("n1q", x)).reshape(B, Hh * Wd, 3, H, hd).permute(2, 0, 3, 1, 4)
            q = qkv.reshape(3, B * H, Hh * Wd, hd)[0]
            qkv = lin(pre + ".qkv", r("n1kv", x)).reshape(B, Hh * Wd, 3, H, hd).permute(2, 0, 3, 1, 4)
            _, k, v = qkv.reshape(3, B * H, Hh * Wd, hd).unbind(0)
            k, v = r("k", k), r("v", v)
            Rh, Rw = rel_tab(Hh, w[pre + ".rel_pos_h"]), rel_tab(Wd, w[pre + ".rel_pos_w"])
            rq = r("qrel", q).reshape(B * H, Hh, Wd, hd)
            q = r("qqk", q)
            rel_h = r("rel", torch.einsum("bhwc,hkc->bhwk", rq, Rh))
            rel_w = r("rel", torch.einsum("bhwc,wkc->bhwk", rq, Rw))
            out = torch.empty(B * H, Hh * Wd, hd, dtype=torch.float64, device=dev)
            qs = r("qs", q * hd ** -0.5)
            step = max(1, (1 << 26) // (Hh * Wd * Hh * Wd))  # heads per chunk (memory)
            for s0 in range(0, B * H, step):
                sl = slice(s0, s0 + step)
                a = qs[sl] @ k[sl].transpose(-2, -1)
                a = (a.view(-1, Hh, Wd, Hh, Wd) + rel_h[sl, :, :, :, None] + rel_w[sl, :, :, None, :]).view(-1, Hh * Wd, Hh * Wd)
                out[sl] = r("p", a.softmax(dim=-1)) @ v[sl]
            o = out.view(B, H, Hh, Wd, hd).permute(0, 2, 3, 1, 4).reshape(B, Hh, Wd, C)
            return lin(pre + ".proj", r("o", o))

        def block(pre, x, ws):
            sc = x
            y = ln(pre + ".norm1", x)
            if ws > 0:
                B, Hh, Wd, C = y.shape
                ph, pw = (ws - Hh % ws) % ws, (ws - Wd % ws) % ws
                y = F.pad(y, (0, 0, 0, pw, 0, ph))
                Hp, Wp = Hh + ph, Wd + pw
                y = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
            y = attention(pre + ".attn", y)
            if ws > 0:
                y = y.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :Hh, :Wd]
            x = sc + y
            h = r("h", F.gelu(lin(pre + ".mlp.lin1", r("n2", ln(pre + ".norm2", x)))))
            return x + lin(pre + ".mlp.lin2", h)

        x = F.conv2d(img, w[p + ".patch_embed.proj.weight"], w[p + ".patch_embed.proj.bias"], stride=c.patch)
        x = x.permute(0, 2, 3, 1) + w[p + ".pos_embed"]
        for i in range(depth):
            x = block(f"{p}.blocks.{i}", x, 0 if i in c.global_attn_indexes else c.window)
        x = F.conv2d(x.permute(0, 3, 1, 2), w[p + ".neck.0.weight"])
        ln2d = lambda pre, t: ((t - t.mean(1, keepdim=True)) / torch.sqrt((t - t.mean(1, keepdim=True)).pow(2).mean(1, keepdim=True) + 1e-6)
                               * w[pre + ".weight"][:, None, None] + w[pre + ".bias"][:, None, None])
        x = ln2d(p + ".neck.1", x)
        return ln2d(p + ".neck.3", F.conv2d(x, w[p + ".neck.2.weight"], padding=1))

    with torch.no_grad():
        ref = run(())
        nrm = float(ref.pow(2).mean().sqrt())
        err = lambda a: float((run(a) - ref).pow(2).mean().sqrt()) / nrm
        print(f"{kind}, depth {depth}: relative rms error of the embedding per rounded site set", flush=True)
        for s in SITES:
            print(f"  {s:4s} {err((s,)):.2e}", flush=True)
        A = ("n1q", "n1kv", "qrel", "qqk", "k", "v", "qs", "p", "o", "n2", "h")  # everything the fp16 mode rounds (fp32 rel-pos terms)
        wo = lambda *x: tuple(t for t in A if t not in x)
        sets = {"all": A, "exact q path (n1q qrel qqk qs)": wo("n1q", "qrel", "qqk", "qs"), "... and p": wo("n1q", "qrel", "qqk", "qs", "p"),
                "exact n1q, q in the rel-pos terms only": wo("n1q", "qrel"), "... and p": wo("n1q", "qrel", "p"),
                "... and p o": wo("n1q", "qrel", "p", "o"), "mlp only (n2 h)": ("n2", "h")}
        for name, a in sets.items():
            print(f"  {name:32s} {err(a):.2e}", flush=True)


if __name__ == "__main__":
    main()
