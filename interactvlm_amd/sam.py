"""SAM ViT image encoder, prompt encoder (text path) and two-way mask decoder on the HIP kernels.

Host-side orchestration only (buffers, views, launch order); every FLOP runs in libivlm_hip.so.
Mirrors model/segment_anything/modeling/{image_encoder,prompt_encoder,mask_decoder,transformer,sam}.py of the
reference.  Activations are kept channels-last ([tokens, C]) end to end: LayerNorm2d becomes a row LayerNorm,
the 1x1 / 3x3 / transposed convolutions become GEMMs, and the decoder consumes the encoder output without the
NCHW<->NHWC permutes the reference performs (transformer.py:82-84, mask_decoder.py:141).
"""
from __future__ import annotations

import math

import torch

from . import ops
from .weights import SAM_PREFIX, SamEncCfg

BF16 = torch.bfloat16
F32 = torch.float32


def _dev(t, device):
    return t.to(device=device, dtype=BF16).contiguous()


class _Lin:
    def __init__(self, w, prefix, device, bias=True):
        self.w = _dev(w[prefix + ".weight"].reshape(w[prefix + ".weight"].shape[0], -1), device)
        self.b = _dev(w[prefix + ".bias"], device) if bias and (prefix + ".bias") in w else None

    def __call__(self, x, act="none", residual=None, res_mod=0, out=None, out_f32=False, out_rows=None, a_rows=None,
                 a_split=False, out_split=False):
        return ops.linear(x, self.w, self.b, act=act, residual=residual, res_mod=res_mod, out=out, out_f32=out_f32,
                          out_rows=out_rows, a_rows=a_rows, a_split=a_split, out_split=out_split)


class _LinF32(_Lin):
    """nn.Linear on fp32 activations through the bf16 matrix cores: the input arrives as [hi | lo] bf16 rows
    (ops.split_rows / the split outputs of add_rows) and meets [W | W] (K' = 2K); bias, activation and an fp32 residual in the
    epilogue, fp32 out.  Used where the FLOPs are negligible and the precision is not (the SAM mask decoder)."""

    def __init__(self, w, prefix, device, bias=True):
        super().__init__(w, prefix, device, bias)
        self.w = torch.cat([self.w, self.w], 1).contiguous()

    def __call__(self, x_split, act="none", residual=None):
        return ops.linear(x_split, self.w, self.b, act=act, residual=residual, out_f32=True)


class _LN:
    def __init__(self, w, prefix, device, eps):
        self.w, self.b, self.eps = _dev(w[prefix + ".weight"], device), _dev(w[prefix + ".bias"], device), eps

    def __call__(self, x, gelu=False, out_f32=False, out=None, out_rows=None, fp8_scale=None, out_split=False, out_f16=False):
        return ops.layernorm(x, self.w, self.b, self.eps, gelu=gelu, out_f32=out_f32, out=out, out_rows=out_rows,
                             fp8_scale=fp8_scale, out_split=out_split, out_f16=out_f16)


# ================================================================================================
# image encoder
# ================================================================================================
class SamImageEncoder:
    """ImageEncoderViT.forward (image_encoder.py:110-125): [V,3,S,S] bf16 -> [V, g*g, 256] bf16 (channels last)."""

    def __init__(self, w, cfg: SamEncCfg, device, prefix=SAM_PREFIX + ".image_encoder"):
        self.cfg, self.device = cfg, device
        p = prefix
        D = cfg.embed_dim
        self.patch = _Lin(w, p + ".patch_embed.proj", device)
        self.pos_embed = _dev(w[p + ".pos_embed"].reshape(-1, D), device)
        self.blocks = []
        for i in range(cfg.depth):
            bp = f"{p}.blocks.{i}"
            self.blocks.append(dict(
                glob=i in cfg.global_attn_indexes,
                norm1=_LN(w, bp + ".norm1", device, 1e-6), norm2=_LN(w, bp + ".norm2", device, 1e-6),
                qkv=_Lin(w, bp + ".attn.qkv", device), proj=_Lin(w, bp + ".attn.proj", device),
                rel_h=_dev(w[bp + ".attn.rel_pos_h"], device), rel_w=_dev(w[bp + ".attn.rel_pos_w"], device),
                lin1=_Lin(w, bp + ".mlp.lin1", device), lin2=_Lin(w, bp + ".mlp.lin2", device)))
        self.neck0 = _Lin(w, p + ".neck.0", device, bias=False)
        self.neck1 = _LN(w, p + ".neck.1", device, 1e-6)
        # conv3x3 weight [O, I, ky, kx] -> GEMM weight [O, (ky, kx, I)] matching im2col3x3_nhwc
        w2 = w[p + ".neck.2.weight"]
        self.neck2_w = _dev(w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1), device)
        self.neck3 = _LN(w, p + ".neck.3", device, 1e-6)
        self._maps = {}
        self._xw = {}  # per view count: the window-ordered q|k|v buffer of the windowed blocks

    def _window_maps(self, V):
        """Row maps of window_partition / window_unpartition (image_encoder.py:263-318) incl. zero padding."""
        if V not in self._maps:
            g, ws = self.cfg.grid, self.cfg.window
            nw = (g + ws - 1) // ws
            gp = nw * ws
            v = torch.arange(V).view(V, 1, 1, 1, 1)
            wy = torch.arange(nw).view(1, nw, 1, 1, 1)
            wx = torch.arange(nw).view(1, 1, nw, 1, 1)
            iy = torch.arange(ws).view(1, 1, 1, ws, 1)
            ix = torch.arange(ws).view(1, 1, 1, 1, ws)
            y, x = wy * ws + iy, wx * ws + ix
            src = (v * g + y) * g + x
            part = torch.where((y < g) & (x < g), src, torch.full_like(src, -1)).reshape(-1)
            yy = torch.arange(g).view(1, g, 1)
            xx = torch.arange(g).view(1, 1, g)
            vv = torch.arange(V).view(V, 1, 1)
            unpart = (((vv * nw + yy // ws) * nw + xx // ws) * ws + yy % ws) * ws + xx % ws
            pad = (part < 0).nonzero().flatten()  # window positions that overhang the grid (zero tokens)
            self._maps[V] = (part.to(torch.int32).to(self.device), unpart.reshape(-1).to(torch.int32).to(self.device),
                             nw, gp, pad.to(torch.int32).to(self.device))
        return self._maps[V]

    # ---- fp8 (OCP e4m3) operands for the four big GEMMs of every block (BASELINE.json configs[4]; opt-in) -------------------
    # Per-tensor scales: the weights are quantised once (amax / 448); the activation scales (norm1 / attention / norm2 / GELU
    # outputs of every block) are calibrated on one bf16 pass over sample images and then FIXED (values beyond them saturate).
    # The quantisation is fused into the producers: the LayerNorms and the mlp1 GEMM's GELU epilogue write e4m3 directly; only
    # the attention output (bf16 kernel) takes one conversion pass.  Patch embedding, attention and neck stay bf16; the residual
    # stream stays fp32.
    fp8 = False
    rel_in_kernel = True  # window attention computes its rel-pos terms itself (ops.attention rel_tab=...); False: relpos kernel
    rel_in_kernel_global = True  # ... and so does the global 64 x 64 attention (bf16 / fp16 operands; the split kernels take arrays)
    parity_window_arrays = True

    def enable_fp8(self, calib_images):
        """calib_images [V,3,S,S]: one bf16 pass records the activation ranges, then the fp8 path is switched on."""
        dev = self.device
        for blk in self.blocks:
            for n in ("qkv", "proj", "lin1", "lin2"):
                q, sc = ops.quantize_fp8(blk[n].w)
                blk[n + "_q"], blk[n + "_s"] = q, sc
            blk["amax"] = {k: torch.zeros(1, dtype=F32, device=dev) for k in ("n1", "att", "n2", "h")}
        self._calibrating = True
        try:
            self._forward(calib_images.to(dev))
        finally:
            self._calibrating = False
        for blk in self.blocks:
            blk["s"] = {k: (v / 448.0).clamp_(min=1e-12) for k, v in blk["amax"].items()}
        self.fp8 = True
        if hasattr(self, "_graphs"):
            self._graphs.clear()

    _calibrating = False

    def _attention(self, blk, xn, V, side, nwin, win=None):
        """Attention.forward (image_encoder.py:235-260) on rows laid out [nwin, side*side, D].  win = (unpart, pad, buffer):
        xn is in IMAGE order and the block is windowed - the q|k|v GEMM runs on the real rows only and scatters them to their
        window positions (window_partition folded into its epilogue); the rows of the padded window positions, whose input is
        zero, are the bias: filled, not computed (16 % of the rows at 64x64 / 14)."""
        c = self.cfg
        H, hd = c.num_heads, c.embed_dim // c.num_heads
        S = side * side
        if win is None:
            qkv = blk["qkv"](xn)  # [nwin*S, 3*D] == [nwin, S, 3, H, hd]
        else:
            unpart, pad, qkv = win
            blk["qkv"](xn, out=qkv, out_rows=unpart)
            ops.fill_rows(qkv, pad, blk["qkv"].b)
        qkv5 = qkv.view(nwin, S, 3, H, hd)
        q, k, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        if self.rel_in_kernel and (2 * side <= 32 or (side == 64 and self.rel_in_kernel_global)) and hd == 80:
            # windows: the decomposed rel-pos terms are computed inside the attention kernel (one small MFMA product per query
            # tile against the [rel_pos_h ; rel_pos_w] table) - no relpos pass, no [B*H, S, 2 side] fp32 arrays
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
            o = ops.attention(q, k, v, hd ** -0.5, rel_tab=(blk["rel_cat"], side))
            return o.permute(0, 2, 1, 3).reshape(nwin * S, H * hd)
        if "rel_cat" not in blk:
            blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
        rel = ops.relpos_bias(q, blk["rel_h"], blk["rel_w"], side, side, cat=blk["rel_cat"])
        o = ops.attention(q, k, v, hd ** -0.5, rel=rel)  # view of a [nwin, S, H, hd] buffer
        return o.permute(0, 2, 1, 3).reshape(nwin * S, H * hd)

    def _block_fp8(self, blk, x, V, nwin, unpart, pad):
        """One block with e4m3 operands for qkv / proj / mlp1 / mlp2 (same dataflow as the bf16 block)."""
        c = self.cfg
        H, hd = c.num_heads, c.embed_dim // c.num_heads
        sc = blk["s"]
        xq = blk["norm1"](x, fp8_scale=sc["n1"])
        glob = blk["glob"]
        side = c.grid if glob else c.window
        S = side * side
        nw_ = V if glob else nwin
        if glob:
            qkv = ops.linear_fp8(xq, blk["qkv_q"], sc["n1"], blk["qkv_s"], blk["qkv"].b)
        else:
            qkv = self._xw[V]
            ops.linear_fp8(xq, blk["qkv_q"], sc["n1"], blk["qkv_s"], blk["qkv"].b, out=qkv, out_rows=unpart)
            ops.fill_rows(qkv, pad, blk["qkv"].b)
        qkv5 = qkv.view(nw_, S, 3, H, hd)
        q, k, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        if self.rel_in_kernel and (2 * side <= 32 or (side == 64 and self.rel_in_kernel_global)) and hd == 80:
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
            a = ops.attention(q, k, v, hd ** -0.5, rel_tab=(blk["rel_cat"], side)).permute(0, 2, 1, 3).reshape(nw_ * S, H * hd)
        else:
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
            rel = ops.relpos_bias(q, blk["rel_h"], blk["rel_w"], side, side, cat=blk["rel_cat"])
            a = ops.attention(q, k, v, hd ** -0.5, rel=rel).permute(0, 2, 1, 3).reshape(nw_ * S, H * hd)
        aq = ops.gather_rows(a, out_kind="fp8", scale=sc["att"])
        x = ops.linear_fp8(aq, blk["proj_q"], sc["att"], blk["proj_s"], blk["proj"].b, residual=x, out=x,
                           a_rows=None if glob else unpart)
        hq = blk["norm2"](x, fp8_scale=sc["n2"])
        h8 = ops.linear_fp8(hq, blk["lin1_q"], sc["n2"], blk["lin1_s"], blk["lin1"].b, act="gelu", out_kind="fp8",
                            scale_out=sc["h"])
        return ops.linear_fp8(h8, blk["lin2_q"], sc["h"], blk["lin2_s"], blk["lin2"].b, residual=x, out=x)

    # The encoder is ~320 launches (ViT-H, 4 views).  Issued one by one they cost the host ~22 ms - during which the language
    # path, launched after it by the same thread, has not even started (measured: the first 21.9 ms of evaluate() had an idle
    # main stream).  Replayed as ONE HIP graph per input shape the host is free after ~20 us.  Same kernels, same order.
    use_graph = True

    def __call__(self, images):
        if not (self.use_graph and images.is_cuda) or torch.cuda.is_current_stream_capturing() or ops.TIMER.enabled:
            return self._forward(images)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        key = tuple(images.shape) + (self.fp8, self.precision, self.parity_sites, self.q_lo_level)
        ent = self._graphs.get(key)
        dev = images.device
        if ent is None:
            static_in = images.to(BF16).contiguous().clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up outside capture (window maps, allocator pools)
                self._forward(static_in)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                static_out = self._forward(static_in)
            ent = self._graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = ent
        static_in.copy_(images)
        g.replay()
        return static_out.clone()

    # ---- "parity" precision (opt-in): fp32-activation arithmetic on the bf16 matrix cores --------------------------------------
    # Every activation that the default mode rounds to a bf16 MFMA operand (normed rows, q / k / v, softmax weights, attention
    # output, MLP hidden) is carried as hi + lo bf16 halves (x = hi + lo to 2^-17): the four GEMMs of a block take [hi | lo]
    # rows against the plain weight (each W tile used twice: 2 x the MFMA work), the attention runs three MFMAs per fragment,
    # rel-pos terms / q scaling / softmax stay fp32.  The bf16 weights are exact in both modes.  Measured on the headline
    # configuration (ViT-H depth 32, 7B): default mode max |dp| 5.8e-3 against the fp32 oracle, this mode < 1e-3 (bench.py).
    precision = "default"

    # which operands travel as hi + lo halves in "parity" precision (diagnostics may switch sites off: tools/diag_precision_modes.py):
    #   n1 / n2: the normed rows (q|k|v and mlp1 GEMM inputs), attn: q, k, v, softmax weights (split attention, fp32 rel-pos terms),
    #   proj: the attention output (proj GEMM input), h: the GELU output (mlp2 GEMM input); rel32 (only without attn): fp32
    #   rel-pos terms in the bf16 attention
    #   f16mlp (instead of n2 / h): the MLP's two GEMMs take fp16 operands - norm2 and the GELU epilogue write IEEE halves (11
    #   significant bits: an eighth of the bf16 rounding error, ONE MFMA pass), the bf16 weights convert to fp16 exactly
    PARITY_SITES = frozenset(("n1", "attn", "proj", "n2", "h"))  # the "parity" mode: nothing below fp32-equivalent operands
    # the encoder of the "parity-encoder" mode: measured 4.6e-4 end to end at depth 32 against 4.0e-4 for PARITY_SITES, 12 ms
    # less per 4 views (tools/diag_precision_modes.py)
    PARITY_SITES_FAST = frozenset(("n1", "attn", "proj", "f16mlp"))
    # the "f16" mode: EVERY MFMA operand of a block as IEEE fp16 in one pass (f16attn: norm1 output, q | k | v, softmax weights,
    # attention output, rel-pos table; f16mlp: norm2 output, GELU hidden) - the bf16 path's launches and FLOPs at an eighth of its
    # operand rounding; the neck (0.1 % of the FLOPs) takes hi + lo operands
    SITES_F16 = frozenset(("f16attn", "f16mlp"))
    # ... with the "exact q" path (f16q): norm1 writes [hi | lo] IEEE halves, q = W_q . (hi + lo) leaves its own GEMM as hi + lo
    # halves (k | v: a single-pass GEMM on the hi half), the attention takes q = hi + lo in the rel-pos table product and in Q.K^T and
    # splits the softmax weights for P.V.  q's rounding is the one SAM's decomposed rel-pos terms amplify (tools/emulate_f16_sites.py:
    # 57 % of the fp16 mode's error variance comes through q), this path removes it for +1/3 of the q|k|v GEMM's MFMA work
    SITES_F16Q = frozenset(("f16attn", "f16q", "f16mlp"))
    q_lo_level = 1
    parity_sites = PARITY_SITES

    def _f16_weights(self, blk, names=("lin1", "lin2")):
        """fp16 copies of a block's GEMM weights.  A bf16 value inside the fp16 NORMAL range (6.1e-5 .. 65504) converts exactly (8
        significant bits into 11); smaller ones land on fp16 subnormals and move by at most 2^-25 = 3e-8 - checked here, and
        irrelevant next to weights of typical size 1e-2."""
        for n in names:
            if n + "_h" not in blk:
                blk[n + "_h"] = ops.f16_weight(blk[n].w, n)
        return tuple(blk[n + "_h"] for n in names)

    def _attention_f16(self, blk, xn, V, side, nwin, win=None, qx=False):
        """_attention on fp16 operands: xn fp16 rows -> fp16 attention output [nwin*S, D].  qx ("exact q"): xn is [rows, 2D] =
        [hi | lo] IEEE halves, win = (unpart, pad, q buffer [.., 2D], k|v buffer [.., 2D])."""
        c = self.cfg
        D = c.embed_dim
        H, hd = c.num_heads, D // c.num_heads
        S = side * side
        (wq,) = self._f16_weights(blk, ("qkv",))
        if "q_b_split_h" not in blk:
            if "qkv_b_h" not in blk:
                blk["qkv_b_h"] = ops.bf16_to_f16(blk["qkv"].b)
            blk["q_b_split_h"] = torch.cat([blk["qkv_b_h"][:D], torch.zeros_like(blk["qkv_b_h"][:D])]).contiguous()
            blk["kv_b_h"] = blk["qkv_b_h"][D:].contiguous()
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
            if "rel_cat_h" not in blk:
                blk["rel_cat_h"] = ops.bf16_to_f16(blk["rel_cat"])
        if qx:
            bq, bkv = blk["qkv"].b[:D], blk["qkv"].b[D:]
            if win is None:
                q2 = ops.linear(xn, wq[:D], bq, a_split=True, out_split=True, out_f16=True)  # [rows, 2D] = [q hi | q lo]
                kv = ops.linear(xn[:, :D], wq[D:], bkv, out_f16=True)                          # [rows, 2D] = [k | v]
            else:
                unpart, pad, q2, kv = win
                ops.linear(xn, wq[:D], bq, out=q2, out_rows=unpart, a_split=True, out_split=True, out_f16=True)
                ops.linear(xn[:, :D], wq[D:], bkv, out=kv, out_rows=unpart)
                ops.fill_rows(q2, pad, blk["q_b_split_h"])
                ops.fill_rows(kv, pad, blk["kv_b_h"])
            q4, kv4 = q2.view(nwin, S, 2, H, hd), kv.view(nwin, S, 2, H, hd)
            q, q_lo, k, v = (t.permute(0, 2, 1, 3) for t in (q4[:, :, 0], q4[:, :, 1], kv4[:, :, 0], kv4[:, :, 1]))
            lv = self.q_lo_level  # 1: q's lo half in the rel-pos terms only (what amplifies its rounding); 2: in Q.K^T too, split P
            # (level 2 on the 64 x 64 grid takes the terms as arrays: its kernel has no table mode)
            if self.rel_in_kernel and (2 * side <= 32 or (side == 64 and self.rel_in_kernel_global and lv < 2)) and hd == 80:
                o = ops.attention(q, k, v, hd ** -0.5, rel_tab=(blk["rel_cat_h"], side), q_lo=q_lo, q_lo_level=lv)
            else:
                rel = ops.relpos_bias(q, blk["rel_h"], blk["rel_w"], side, side, cat=blk["rel_cat_h"], q_lo=q_lo)
                o = ops.attention(q, k, v, hd ** -0.5, rel=rel, q_lo=q_lo if lv == 2 else None, q_lo_level=lv)
            return o.permute(0, 2, 1, 3).reshape(nwin * S, H * hd)
        if win is None:
            qkv = ops.linear(xn, wq, blk["qkv"].b, out_f16=True)
        else:
            unpart, pad, qkv = win
            ops.linear(xn, wq, blk["qkv"].b, out=qkv, out_rows=unpart)
            ops.fill_rows(qkv, pad, blk["qkv_b_h"])
        qkv5 = qkv.view(nwin, S, 3, H, hd)
        q, k, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        if self.rel_in_kernel and (2 * side <= 32 or (side == 64 and self.rel_in_kernel_global)) and hd == 80:
            o = ops.attention(q, k, v, hd ** -0.5, rel_tab=(blk["rel_cat_h"], side))
        else:
            rel = ops.relpos_bias(q, blk["rel_h"], blk["rel_w"], side, side, cat=blk["rel_cat_h"])
            o = ops.attention(q, k, v, hd ** -0.5, rel=rel)
        return o.permute(0, 2, 1, 3).reshape(nwin * S, H * hd)

    def _attention_parity(self, blk, xn, V, side, nwin, win=None):
        """_attention with split operands: xn [rows, D or 2D] -> attention output [nwin*S, 2D] ([hi | lo] rows), or
        [nwin*S, D] bf16 when the 'attn' site is off."""
        c = self.cfg
        D = c.embed_dim
        H, hd = c.num_heads, D // c.num_heads
        S = side * side
        sites = self.parity_sites
        sa, sn = "attn" in sites, "n1" in sites
        if win is None:
            qkv = blk["qkv"](xn, a_split=sn, out_split=sa)  # [nwin*S, (2 *) 3D]
        else:
            unpart, pad, qkv = win
            blk["qkv"](xn, out=qkv, out_rows=unpart, a_split=sn, out_split=sa)
            if sa and "qkv_b_split" not in blk:
                blk["qkv_b_split"] = torch.cat([blk["qkv"].b, torch.zeros_like(blk["qkv"].b)]).contiguous()
            ops.fill_rows(qkv, pad, blk["qkv_b_split"] if sa else blk["qkv"].b)
        if not sa:
            qkv5 = qkv.view(nwin, S, 3, H, hd)
            q, k, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            if "rel32" in sites:
                rel = ops.relpos_bias_split(q, None, blk["rel_h"], blk["rel_w"], side, side)
            else:
                if "rel_cat" not in blk:
                    blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
                rel = ops.relpos_bias(q, blk["rel_h"], blk["rel_w"], side, side, cat=blk["rel_cat"])
            return ops.attention(q, k, v, hd ** -0.5, rel=rel).permute(0, 2, 1, 3).reshape(nwin * S, H * hd)
        q6 = qkv.view(nwin, S, 2, 3, H, hd)
        hi = [q6[:, :, 0, i].permute(0, 2, 1, 3) for i in range(3)]
        lo = [q6[:, :, 1, i].permute(0, 2, 1, 3) for i in range(3)]
        # windows: the whole-window split kernel has no LDS left for the table product, it takes the fp32 terms as arrays
        # (parity_window_arrays; measured faster than the generic split kernel in table mode)
        if self.rel_in_kernel and not self.parity_window_arrays and 2 * side <= 32 and hd == 80:
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
            return ops.attention_split(hi[0], lo[0], hi[1], lo[1], hi[2], lo[2], hd ** -0.5, rel_tab=(blk["rel_cat"], side))
        rel = ops.relpos_bias_split(hi[0], lo[0], blk["rel_h"], blk["rel_w"], side, side)
        return ops.attention_split(hi[0], lo[0], hi[1], lo[1], hi[2], lo[2], hd ** -0.5, rel=rel)  # [nwin*S, 2D]

    def _forward_parity(self, images):
        c = self.cfg
        V = images.shape[0]
        g, D = c.grid, c.embed_dim
        sites = self.parity_sites
        sa = "attn" in sites
        sp = sa and "proj" in sites
        cols = ops.im2col_nchw(images.to(BF16).contiguous(), c.patch, c.patch)  # (bf16 pixels x bf16 weights: exact products)
        x = self.patch(cols, residual=self.pos_embed, res_mod=g * g, out_f32=True)
        part, unpart, nw, gp, pad = self._window_maps(V)
        nwin = V * nw * nw
        f16a = "f16attn" in sites
        key = ("f16" if f16a else ("split" if sa else "bf16"), V)
        if key not in self._xw:
            self._xw[key] = torch.empty(nwin * c.window * c.window, (6 if sa else 3) * D, dtype=torch.float16 if f16a else BF16,
                                        device=x.device)
        qx = f16a and "f16q" in sites
        if qx and ("f16q", V) not in self._xw:  # window-ordered [q hi | q lo] and [k | v] buffers
            self._xw[("f16q", V)] = tuple(torch.empty(nwin * c.window * c.window, 2 * D, dtype=torch.float16, device=x.device)
                                          for _ in range(2))
        for blk in self.blocks:
            if f16a:
                xn = blk["norm1"](x, out_f16=True, out_split=qx)
                (wp,) = self._f16_weights(blk, ("proj",))
                if blk["glob"]:
                    a = self._attention_f16(blk, xn, V, g, V, qx=qx)
                    x = ops.linear(a, wp, blk["proj"].b, residual=x, out_f32=True)
                else:
                    a = self._attention_f16(blk, xn, V, c.window, nwin, qx=qx,
                                            win=(unpart, pad) + (self._xw[("f16q", V)] if qx else (self._xw[key],)))
                    x = ops.linear(a, wp, blk["proj"].b, residual=x, out=x, a_rows=unpart)
            elif blk["glob"]:
                a = self._attention_parity(blk, blk["norm1"](x, out_split="n1" in sites), V, g, V)
                x = blk["proj"](a if (sp or not sa) else a[:, :D], residual=x, out_f32=True, a_split=sp)
            else:
                a = self._attention_parity(blk, blk["norm1"](x, out_split="n1" in sites), V, c.window, nwin,
                                           win=(unpart, pad, self._xw[key]))
                x = blk["proj"](a if (sp or not sa) else a[:, :D], residual=x, out=x, a_rows=unpart, a_split=sp)
            if "f16mlp" in sites:
                w1, w2 = self._f16_weights(blk)
                h = ops.linear(blk["norm2"](x, out_f16=True), w1, blk["lin1"].b, act="gelu", out_f16=True)
                x = ops.linear(h, w2, blk["lin2"].b, residual=x, out_f32=True)
                continue
            h = blk["lin1"](blk["norm2"](x, out_split="n2" in sites), act="gelu", a_split="n2" in sites, out_split="h" in sites)
            x = blk["lin2"](h, residual=x, out_f32=True, a_split="h" in sites)
        y = self.neck0(ops.gather_rows(x, out_kind="split"), out_f32=True, a_split=True)
        y = self.neck1(y, out_split=True)  # [V*g*g, 2 * 256]
        y = ops.linear(ops.im2col3x3_nhwc_split(y, V, g, g, c.out_chans), self.neck2_w, out_f32=True, a_split=True)
        return self.neck3(y, out_f32=True).view(V, g * g, c.out_chans)

    def _forward(self, images):
        """The residual stream x is fp32 (GEMM residual epilogues write it, the LayerNorms read it); MFMA operands are bf16.
        window_partition is folded into the q|k|v GEMM's scatter epilogue and window_unpartition + shortcut into the proj GEMM's
        gather prologue: both GEMMs of a windowed block run on the g*g real rows of every view only - no gather passes over the
        activations, no work on the padded window positions (whose q|k|v rows are just the bias)."""
        if self.precision == "parity":
            return self._forward_parity(images)
        c = self.cfg
        V = images.shape[0]
        g, D = c.grid, c.embed_dim
        cols = ops.im2col_nchw(images.to(BF16).contiguous(), c.patch, c.patch)
        x = self.patch(cols, residual=self.pos_embed, res_mod=g * g, out_f32=True)  # + pos_embed broadcast over views
        part, unpart, nw, gp, pad = self._window_maps(V)
        nwin = V * nw * nw
        if V not in self._xw:  # q|k|v in window order (one buffer for all windowed blocks of this view count)
            self._xw[V] = torch.empty(nwin * c.window * c.window, 3 * D, dtype=BF16, device=x.device)
        for blk in self.blocks:
            if self.fp8:
                x = self._block_fp8(blk, x, V, nwin, unpart, pad)
                continue
            cal = blk.get("amax") if self._calibrating else None
            xn = blk["norm1"](x)
            if cal:
                ops.amax(xn, cal["n1"])
            if blk["glob"]:
                a = self._attention(blk, xn, V, g, V)
                if cal:
                    ops.amax(a, cal["att"])
                x = blk["proj"](a, residual=x, out_f32=True)
            else:
                a = self._attention(blk, xn, V, c.window, nwin, win=(unpart, pad, self._xw[V]))
                if cal:
                    ops.amax(a.contiguous(), cal["att"])
                # proj + window_unpartition + shortcut, in place: the GEMM runs on the g*g real rows of every view only (its A
                # rows are gathered from their window positions; the rows of the padded window grid are never computed)
                x = blk["proj"](a, residual=x, out=x, a_rows=unpart)
            xn2 = blk["norm2"](x)
            h = blk["lin1"](xn2, act="gelu")
            if cal:
                ops.amax(xn2, cal["n2"])
                ops.amax(h, cal["h"])
            x = blk["lin2"](h, residual=x, out_f32=True)
        y = self.neck1(self.neck0(ops.gather_rows(x, out_kind="bf16")))
        y = ops.linear(ops.im2col3x3_nhwc(y.view(V, g, g, c.out_chans)), self.neck2_w)
        return self.neck3(y, out_f32=True).view(V, g * g, c.out_chans)  # fp32: the mask decoder keeps fp32 activations


# ================================================================================================
# prompt encoder (text-embedding path) + mask decoder + postprocess
# ================================================================================================
class SamMaskDecoder:
    """PromptEncoder.forward(text_embeds=...) + MaskDecoder.forward(multimask_output=False)
    (prompt_encoder.py:140-186, mask_decoder.py:75-164, transformer.py:62-242).

    Precision: this stage is 15 GFLOP of the image's 27 TFLOP and it writes the mask logits, so it runs with fp32 activations
    end to end: every nn.Linear / transposed conv takes its input as hi + lo bf16 rows against [W | W] (``_LinF32``: an
    fp32-activation GEMM on the bf16 matrix cores, weights exactly the checkpoint's bf16), the attentions run in fp32
    (``ops.attention_f32``), LayerNorms and the hypernetwork product read and write fp32."""

    def __init__(self, w, device, grid=64, prefix=SAM_PREFIX, decoder="mask_decoder"):
        """decoder: the attribute name of the decoder module to load ('-DifDe' checkpoints also carry 'human_mask_decoder' and
        'object_mask_decoder', separately trained copies: InteractVLM.py:114-121); the prompt encoder is shared."""
        self.device, self.grid = device, grid
        pe, md = prefix + ".prompt_encoder", prefix + "." + decoder
        self.C = C = w[md + ".iou_token.weight"].shape[1]
        f32 = lambda t: t.to(device=device, dtype=BF16).to(F32).contiguous()  # the checkpoint's bf16 values, held as fp32
        self.no_mask = f32(w[pe + ".no_mask_embed.weight"].reshape(1, C))
        gauss = w[pe + ".pe_layer.positional_encoding_gaussian_matrix"].to(device=device, dtype=torch.float32).contiguous()
        self.key_pe = ops.dense_pe(gauss, grid, grid)  # fp32 [g*g, C] constant: computed once, not per call
        self.out_tokens = f32(torch.cat([w[md + ".iou_token.weight"], w[md + ".mask_tokens.weight"]], 0))
        self.n_mask = w[md + ".mask_tokens.weight"].shape[0]
        tp = md + ".transformer"

        def attn(p):
            return dict(q=_LinF32(w, p + ".q_proj", device), k=_LinF32(w, p + ".k_proj", device),
                        v=_LinF32(w, p + ".v_proj", device), o=_LinF32(w, p + ".out_proj", device))

        self.layers = []
        i = 0
        while f"{tp}.layers.{i}.norm1.weight" in w:
            lp = f"{tp}.layers.{i}"
            self.layers.append(dict(
                self_attn=attn(lp + ".self_attn"), t2i=attn(lp + ".cross_attn_token_to_image"),
                i2t=attn(lp + ".cross_attn_image_to_token"),
                norm1=_LN(w, lp + ".norm1", device, 1e-5), norm2=_LN(w, lp + ".norm2", device, 1e-5),
                norm3=_LN(w, lp + ".norm3", device, 1e-5), norm4=_LN(w, lp + ".norm4", device, 1e-5),
                lin1=_LinF32(w, lp + ".mlp.lin1", device), lin2=_LinF32(w, lp + ".mlp.lin2", device)))
            i += 1
        self.final_attn = attn(tp + ".final_attn_token_to_image")
        self.norm_final = _LN(w, tp + ".norm_final_attn", device, 1e-5)
        # ConvTranspose2d(k=2,s=2) as GEMM: weight [ci, co, dy, dx] -> [(dy, dx, co), ci]; bias tiled over (dy,dx)
        w0 = w[md + ".output_upscaling.0.weight"]
        u0 = _dev(w0.permute(2, 3, 1, 0).reshape(-1, w0.shape[0]), device)
        self.up0_w = torch.cat([u0, u0], 1).contiguous()
        self.up0_b = _dev(w[md + ".output_upscaling.0.bias"].repeat(4), device)
        self.up_ln = _LN(w, md + ".output_upscaling.1", device, 1e-6)
        w1 = w[md + ".output_upscaling.3.weight"]
        u1 = _dev(w1.permute(2, 3, 1, 0).reshape(-1, w1.shape[0]), device)
        self.up1_w = torch.cat([u1, u1], 1).contiguous()
        self.up1_b = _dev(w[md + ".output_upscaling.3.bias"].repeat(4), device)
        self.c_mid, self.c_up = w0.shape[1], w1.shape[1]
        self.hyper0 = [_LinF32(w, f"{md}.output_hypernetworks_mlps.0.layers.{j}", device) for j in range(3)]
        self.iou = [_LinF32(w, f"{md}.iou_prediction_head.layers.{j}", device) for j in range(3)]

    def _attn(self, a, q_split, k_split, v_split, B, Sq, Sk, heads=8, kv_batch=None):
        """Attention.forward (transformer.py:220-242). *_split are [B*S, 2C] hi|lo rows -> fp32 [B*Sq, inner] (before out_proj)."""
        q, k, v = a["q"](q_split), a["k"](k_split), a["v"](v_split)
        inner = q.shape[-1]
        d = inner // heads
        Bk = B if kv_batch is None else kv_batch
        q4 = q.view(B, Sq, heads, d).permute(0, 2, 1, 3)
        k4 = k.view(Bk, Sk, heads, d).permute(0, 2, 1, 3)
        v4 = v.view(Bk, Sk, heads, d).permute(0, 2, 1, 3)
        o = ops.attention_f32(q4, k4, v4, 1.0 / math.sqrt(d))
        return ops.split_rows(o.permute(0, 2, 1, 3).reshape(B * Sq, inner))

    # The decoder chain (prompt tokens -> two-way transformer -> upscaler -> hypernetwork dot -> IoU head) is ~200 launches of
    # 3-30 us kernels with no host decision inside: replayed as ONE HIP graph per (views, tokens) shape (BASELINE.json
    # configs[4]: "fused SAM decoder in one hipGraph").  Same kernels, same order: bit-identical to the eager chain.
    use_graph = True

    def __call__(self, image_embeddings, text_embeds):
        if (not (self.use_graph and image_embeddings.is_cuda) or torch.cuda.is_current_stream_capturing()
                or ops.TIMER.enabled):
            return self._forward(image_embeddings, text_embeds)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        key = (tuple(image_embeddings.shape), image_embeddings.dtype, tuple(text_embeds.shape), text_embeds.dtype)
        ent = self._graphs.get(key)
        dev = image_embeddings.device
        if ent is None:
            s_emb, s_txt = image_embeddings.contiguous().clone(), text_embeds.contiguous().clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up outside capture (allocator pools, lazy module loads)
                self._forward(s_emb, s_txt)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                s_out = self._forward(s_emb, s_txt)
            ent = self._graphs[key] = (g, s_emb, s_txt, s_out)
        g, s_emb, s_txt, s_out = ent
        s_emb.copy_(image_embeddings)
        s_txt.copy_(text_embeds)
        g.replay()
        return s_out[0].clone(), s_out[1].clone()

    def _forward(self, image_embeddings, text_embeds):
        """image_embeddings [V, g*g, C] fp32 or bf16 (channels last); text_embeds [1, T, C] (the views as TOKENS)
        -> low_res_masks f32 [V,1,4g,4g], iou f32 [V,1].

        Batch semantics follow torch broadcasting in the reference exactly: one token set of
        5+T tokens; the first self-attention sees batch 1, every later op batch V.  We carry V identical copies
        from the start (same numbers) so that every kernel sees a fixed batch of V."""
        V, HW, C = image_embeddings.shape
        g = self.grid
        assert text_embeds.shape[0] == 1, "n_seg > 1 with multi-view mis-broadcasts in the reference (SURVEY §7)"
        tokens = torch.cat([self.out_tokens, text_embeds[0].to(F32)], dim=0)  # [Nt, C]
        Nt = tokens.shape[0]
        query_pe = tokens.unsqueeze(0).expand(V, Nt, C).reshape(V * Nt, C).contiguous()
        queries = query_pe
        keys = ops.add_rows(image_embeddings.reshape(V * HW, C).contiguous(), self.no_mask, out_kind="f32")  # src + dense
        key_pe = self.key_pe  # [HW, C], broadcast over V by row modulo
        sp = ops.split_rows
        for li, L in enumerate(self.layers):
            if li == 0:  # skip_first_layer_pe: queries = self_attn(q=k=v=queries), no residual
                qs = sp(queries)
                queries = L["self_attn"]["o"](self._attn(L["self_attn"], qs, qs, qs, V, Nt, Nt))
            else:
                q = ops.add_rows(queries, query_pe, out_kind="split")
                sa = self._attn(L["self_attn"], q, q, sp(queries), V, Nt, Nt)
                queries = L["self_attn"]["o"](sa, residual=queries)
            queries = L["norm1"](queries, out_f32=True)
            q = ops.add_rows(queries, query_pe, out_kind="split")
            k = ops.add_rows(keys, key_pe, out_kind="split")
            ca = self._attn(L["t2i"], q, k, sp(keys), V, Nt, HW)
            queries = L["norm2"](L["t2i"]["o"](ca, residual=queries), out_f32=True)
            mlp = L["lin2"](sp(L["lin1"](sp(queries), act="relu")), residual=queries)
            queries = L["norm3"](mlp, out_f32=True)
            q = ops.add_rows(queries, query_pe, out_kind="split")
            ia = self._attn(L["i2t"], k, q, sp(queries), V, HW, Nt)  # image attends to tokens (q=k_img, k=q_tok)
            keys = L["norm4"](L["i2t"]["o"](ia, residual=keys), out_f32=True)
        q = ops.add_rows(queries, query_pe, out_kind="split")
        k = ops.add_rows(keys, key_pe, out_kind="split")
        fa = self._attn(self.final_attn, q, k, sp(keys), V, Nt, HW)
        hs = self.norm_final(self.final_attn["o"](fa, residual=queries), out_f32=True).view(V, Nt, C)
        iou_tok = hs[:, 0, :].contiguous()
        mask_tok0 = hs[:, 1, :].contiguous()  # mask token 0: multimask_output=False keeps masks[:, 0:1]
        # output_upscaling: ConvT(256->64) -> LayerNorm2d -> GELU -> ConvT(64->32) -> GELU, as GEMMs on pixels
        u = ops.linear(sp(keys), self.up0_w, self.up0_b, out_f32=True)  # [V*HW, (dy,dx,64)]
        u = self.up_ln(u.view(-1, self.c_mid), gelu=True, out_f32=True)  # per output pixel over 64 channels
        u = ops.linear(sp(u), self.up1_w, self.up1_b, act="gelu", out_f32=True)  # [V*HW*4, (dy2,dx2,32)]
        h = self.hyper0[2](sp(self.hyper0[1](sp(self.hyper0[0](sp(mask_tok0), act="relu")), act="relu")))  # [V, 32]
        low = ops.mask_dot(u, h, V, g, g)  # f32 [V, 4g, 4g]
        iou = self.iou[2](sp(self.iou[1](sp(self.iou[0](sp(iou_tok), act="relu")), act="relu")))
        return low.unsqueeze(1), iou[:, 0:1]


def postprocess_masks(low_res, input_size, original_size, img_size=1024, apply_sigmoid=False, sigmoid_gt=None, ignore_label=-1.0):
    """Sam.postprocess_masks (sam.py:137-172) (+ the masked sigmoid of InteractVLM.py:452-456 when sigmoid_gt is given)."""
    return ops.postprocess_masks(low_res.contiguous(), input_size, original_size, img_size, apply_sigmoid, sigmoid_gt, ignore_label)
