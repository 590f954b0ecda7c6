"""CLIP ViT-L/14 tower + mm_projector + LLaMA decoder (KV-cached) on the HIP kernels.

Mirrors the arithmetic the reference gets from HF ``CLIPVisionModel`` / ``LlamaModel`` through
model/llava/model/{llava_arch.py, language_model/llava_llama.py, multimodal_encoder/clip_encoder.py}.
MI355X-first differences (same numbers, less work):
  * CLIP is encoded ONCE per image; the reference re-encodes it for every generated token
    (InteractVLM.py:128 use_cache=False -> llava_arch.py:98-123 on every step);
  * decode uses a KV cache (the reference re-runs the full prefix each step);
  * q/k/v and gate/up projections are fused GEMMs over concatenated / row-interleaved weights, SwiGLU and
    residual adds live in GEMM epilogues, image features are written straight into the LLM input buffer.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from .sam import _LN, _Lin, _dev
from .weights import CLIP_PREFIX, ClipCfg, LlamaCfg

BF16 = torch.bfloat16
F32 = torch.float32


class ClipTower:
    """CLIPVisionTower.forward + feature_select('patch', layer -2) (clip_encoder.py:31-60)."""

    def __init__(self, w, cfg: ClipCfg, device, prefix=CLIP_PREFIX):
        self.cfg, self.device = cfg, device
        self._graphs = {}
        self._patch_rows = {}
        e = prefix + ".embeddings"
        K = 3 * cfg.patch * cfg.patch
        self.kpad = ((K + 63) // 64) * 64
        pw = w[e + ".patch_embedding.weight"].reshape(cfg.hidden, K)
        self.patch_w = _dev(torch.nn.functional.pad(pw, (0, self.kpad - K)), device)
        pos = w[e + ".position_embedding.weight"]
        self.pos = _dev(pos, device)
        self.cls_row = (w[e + ".class_embedding"].float() + pos[0].float()).reshape(1, -1).to(device).contiguous()  # fp32
        self.pre_ln = _LN(w, prefix + ".pre_layrnorm", device, cfg.eps)
        n_run = cfg.layers + 1 + cfg.select_layer if cfg.select_layer < 0 else cfg.select_layer
        self.layers = []
        for i in range(n_run):  # hidden_states[-2] is the output of layer L-1: the last layer is never needed
            p = f"{prefix}.encoder.layers.{i}"
            qkv_w = torch.cat([w[f"{p}.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
            qkv_b = torch.cat([w[f"{p}.self_attn.{n}_proj.bias"] for n in "qkv"], 0)
            self.layers.append(dict(
                ln1=_LN(w, p + ".layer_norm1", device, cfg.eps), ln2=_LN(w, p + ".layer_norm2", device, cfg.eps),
                qkv_w=_dev(qkv_w, device), qkv_b=_dev(qkv_b, device), out=_Lin(w, p + ".self_attn.out_proj", device),
                fc1=_Lin(w, p + ".mlp.fc1", device), fc2=_Lin(w, p + ".mlp.fc2", device)))

    use_graph = True  # replay the tower as one HIP graph (it is ~250 launches of 5-15 us kernels: launch-bound)

    def __call__(self, images):
        """images [B,3,S,S] bf16 -> patch features [B, T-1, hidden]."""
        if not (self.use_graph and images.is_cuda) or torch.cuda.is_current_stream_capturing() or ops.TIMER.enabled:
            return self._forward(images)
        B = images.shape[0]
        gkey = (B, self.precision, self.fp8)
        ent = self._graphs.get(gkey)
        if ent is None:
            static_in = images.to(BF16).contiguous().clone()
            side = torch.cuda.Stream(device=images.device)
            side.wait_stream(torch.cuda.current_stream(images.device))
            with torch.cuda.stream(side):  # warm-up outside capture (first-use attribute calls, allocator pools)
                self._forward(static_in)
            torch.cuda.current_stream(images.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (an RCCL watchdog thread may be polling events)
                static_out = self._forward(static_in)
            ent = self._graphs[gkey] = (g, static_in, static_out)
        g, static_in, static_out = ent
        static_in.copy_(images)
        g.replay()
        return static_out.clone()

    # "parity": fp32-activation arithmetic (hi + lo bf16 operands), see SamImageEncoder.precision; "f16": every MFMA operand as
    # IEEE fp16 (one pass, an eighth of the bf16 operand rounding; fp16 copies of the weights made on first use)
    precision = "default"

    def _f16(self, L):
        if "qkv_h" not in L:
            L["qkv_h"] = ops.f16_weight(L["qkv_w"], "clip qkv")
            for n in ("out", "fc1", "fc2"):
                L[n + "_h"] = ops.f16_weight(L[n].w, "clip " + n)
        return L

    # ---- fp8 (OCP e4m3) operands for the four GEMMs of every layer (BASELINE.json configs[4]; opt-in): per-tensor scales, the
    # weights quantised once, the activation ranges calibrated on one bf16 pass over calibration images and then fixed.
    fp8 = False
    _calibrating = False

    def enable_fp8(self, calib_images):
        dev = self.device
        for L in self.layers:
            for n, w_ in (("qkv", L["qkv_w"]), ("out", L["out"].w), ("fc1", L["fc1"].w), ("fc2", L["fc2"].w)):
                L[n + "_q"], L[n + "_s"] = ops.quantize_fp8(w_)
            L["amax"] = {k: torch.zeros(1, dtype=F32, device=dev) for k in ("y1", "a", "y2", "h")}
        self._calibrating = True
        try:
            self._forward(calib_images.to(dev))
        finally:
            self._calibrating = False
        for L in self.layers:
            L["s"] = {k: (v / 448.0).clamp_(min=1e-12) for k, v in L["amax"].items()}
        self.fp8 = True
        self._graphs.clear()

    def _forward(self, images):
        """-> penultimate-layer patch features, bf16 [B, T-1, hidden] (the mm_projector's MFMA operand; "parity" precision:
        [B, T-1, 2*hidden] = [hi | lo] rows).  The residual stream is fp32 between the GEMMs (residual epilogues write fp32,
        the LayerNorms read it)."""
        par = self.precision == "parity"
        c = self.cfg
        B = images.shape[0]
        T, Hh, hd = c.tokens, c.heads, c.hidden // c.heads
        cols = ops.im2col_nchw(images.to(BF16).contiguous(), c.patch, c.patch, self.kpad)  # [B*(T-1), kpad]
        x = torch.empty(B, T, c.hidden, dtype=F32, device=images.device)
        for b in range(B):  # patch GEMM writes rows 1..T-1 and adds their position embeddings in the epilogue
            ops.linear(cols[b * (T - 1): (b + 1) * (T - 1)], self.patch_w, residual=self.pos[1:], out=x[b, 1:])
            ops.gather_rows(self.cls_row, out=x[b, 0:1])  # class_embedding + position_embedding[0] (precomputed constant)
        x = self.pre_ln(x.view(B * T, c.hidden), out_f32=True)
        for L in self.layers:
            if self.fp8:  # e4m3 operands for qkv / out / fc1 / fc2 (attention stays bf16, the residual stream fp32)
                sc = L["s"]
                qkv = ops.linear_fp8(L["ln1"](x, fp8_scale=sc["y1"]), L["qkv_q"], sc["y1"], L["qkv_s"], L["qkv_b"]).view(B, T, 3, Hh, hd)
                q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
                a = ops.attention(q, k, v, hd ** -0.5, prescale_q=True).permute(0, 2, 1, 3).reshape(B * T, c.hidden)
                aq = ops.gather_rows(a, out_kind="fp8", scale=sc["a"])
                x = ops.linear_fp8(aq, L["out_q"], sc["a"], L["out_s"], L["out"].b, residual=x, out_kind="f32")
                h8 = ops.linear_fp8(L["ln2"](x, fp8_scale=sc["y2"]), L["fc1_q"], sc["y2"], L["fc1_s"], L["fc1"].b, act="quick_gelu",
                                    out_kind="fp8", scale_out=sc["h"])
                x = ops.linear_fp8(h8, L["fc2_q"], sc["h"], L["fc2_s"], L["fc2"].b, residual=x, out_kind="f32")
                continue
            if par:  # every MFMA operand as hi + lo halves; q * scale, softmax in fp32
                qkv = ops.linear(L["ln1"](x, out_split=True), L["qkv_w"], L["qkv_b"], a_split=True, out_split=True)
                q6 = qkv.view(B, T, 2, 3, Hh, hd)
                hi = [q6[:, :, 0, i].permute(0, 2, 1, 3) for i in range(3)]
                lo = [q6[:, :, 1, i].permute(0, 2, 1, 3) for i in range(3)]
                a = ops.attention_split(hi[0], lo[0], hi[1], lo[1], hi[2], lo[2], hd ** -0.5, prescale_q=True)
                x = L["out"](a, residual=x, out_f32=True, a_split=True)
                h = L["fc1"](L["ln2"](x, out_split=True), act="quick_gelu", a_split=True, out_split=True)
                x = L["fc2"](h, residual=x, out_f32=True, a_split=True)
                continue
            if self.precision == "f16":
                L = self._f16(L)
                qkv = ops.linear(L["ln1"](x, out_f16=True), L["qkv_h"], L["qkv_b"], out_f16=True).view(B, T, 3, Hh, hd)
                q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
                a = ops.attention(q, k, v, hd ** -0.5, prescale_q=True).permute(0, 2, 1, 3).reshape(B * T, c.hidden)
                x = ops.linear(a, L["out_h"], L["out"].b, residual=x, out_f32=True)
                h = ops.linear(L["ln2"](x, out_f16=True), L["fc1_h"], L["fc1"].b, act="quick_gelu", out_f16=True)
                x = ops.linear(h, L["fc2_h"], L["fc2"].b, residual=x, out_f32=True)
                continue
            cal = L.get("amax") if self._calibrating else None
            y = L["ln1"](x)
            qkv = ops.linear(y, L["qkv_w"], L["qkv_b"]).view(B, T, 3, Hh, hd)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            a = ops.attention(q, k, v, hd ** -0.5, prescale_q=True).permute(0, 2, 1, 3).reshape(B * T, c.hidden)
            x = L["out"](a, residual=x, out_f32=True)
            y2 = L["ln2"](x)
            h = L["fc1"](y2, act="quick_gelu")
            if cal:
                for k_, t_ in (("y1", y), ("a", a), ("y2", y2), ("h", h)):
                    ops.amax(t_, cal[k_])
            x = L["fc2"](h, residual=x, out_f32=True)
        if B not in self._patch_rows:  # drop the CLS row of every image
            r = torch.arange(B * T, dtype=torch.int32).view(B, T)[:, 1:].reshape(-1)
            self._patch_rows[B] = r.to(images.device)
        if par or self.precision == "f16":  # (the mm_projector takes hi + lo rows in both: 0.03 % of the image's FLOPs)
            return ops.gather_rows(x, self._patch_rows[B], out_kind="split").view(B, T - 1, 2 * c.hidden)
        return ops.gather_rows(x, self._patch_rows[B], out_kind="bf16").view(B, T - 1, c.hidden)


class _LayerW(dict):
    """One LLaMA layer's tensors: ln1 / ln2, the four bf16 matrices (qkv, o, gu, down) and their derived copies (``*_h``: IEEE fp16
    for the fp16-operand prefill, ``*_p``: ops.PackedBf12 planes for the decode step).  A bf16 matrix that ``Llama.release_unused``
    dropped comes back on access, bit for bit, from its packed planes (``ivlm_unpack_bf12``: the 12-bit format is lossless)."""

    MATS = ("qkv", "o", "gu", "down")

    def __missing__(self, k):
        p = dict.get(self, k + "_p") if k in self.MATS else None
        if p is None:
            raise KeyError(k)
        w = p.unpack().contiguous()
        self[k] = w
        return w


class Llama:
    """HF LlamaModel + lm_head with a KV cache, one sequence (batch 1) per instance call."""

    MAX_LEN_LIMIT = 4096  # the decode attention kernel keeps one score per cached position in LDS (decattn::kMaxT)

    def __init__(self, w, cfg: LlamaCfg, device, prefix="model", max_len=640):
        if max_len > self.MAX_LEN_LIMIT:
            raise ops.IvlmError(f"max_len={max_len}: the KV-cached decode kernels support at most {self.MAX_LEN_LIMIT} positions")
        self.cfg, self.device, self.max_len = cfg, device, max_len
        self.embed = _dev(w[prefix + ".embed_tokens.weight"], device)
        self.layers = []
        for i in range(cfg.layers):
            p = f"{prefix}.layers.{i}"
            qkv = torch.cat([w[f"{p}.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
            gu = torch.stack([w[p + ".mlp.gate_proj.weight"], w[p + ".mlp.up_proj.weight"]], 1).reshape(
                2 * cfg.inter, cfg.hidden)  # rows (gate_j, up_j) interleaved for the SwiGLU epilogue
            self.layers.append(_LayerW(
                ln1=_dev(w[p + ".input_layernorm.weight"], device), ln2=_dev(w[p + ".post_attention_layernorm.weight"], device),
                qkv=_dev(qkv, device), o=_dev(w[p + ".self_attn.o_proj.weight"], device), gu=_dev(gu, device),
                down=_dev(w[p + ".mlp.down_proj.weight"], device)))
        self.norm = _dev(w[prefix + ".norm.weight"], device)
        self.lm_head = _dev(w["lm_head.weight"], device)
        H, hd = cfg.heads, cfg.hidden // cfg.heads
        self.kcache = torch.zeros(cfg.layers, max_len, H, hd, dtype=BF16, device=device)
        self.vcache = torch.zeros(cfg.layers, max_len, H, hd, dtype=BF16, device=device)
        self.rope = ops.rope_table(max_len, hd, cfg.theta, device)  # fp32 cos/sin, computed once
        self.precision = "default"
        self.decode_splitkv = False  # split-KV decode attention (long prompts): see _attn_scratch
        self.kcache_lo = self.vcache_lo = None  # "parity" precision: lo planes of the cache (allocated on first use)
        self._dgraphs = {}
        self._dgraph = None
        self._fused = None
        # attention + o_proj in ONE launch (o_proj blocks wait on device counters): saved a launch per layer when the o_proj
        # GEMV was the persistent kernel; with gemv1_kernel the separate launches are as fast (2.67 vs 2.68 ms/token, same
        # end to end), so the simpler graph is the default and the fused launch stays opt-in (tests cover both)
        self.fuse_attn_oproj = bool(os.environ.get("IVLM_FUSE_ATTN_OPROJ"))
        # The batch-1 decode linears stream LOSSLESSLY packed weights (ops.PackedBf12: 1.5 bytes per weight, every bf16 value
        # reconstructed bit for bit, dots on the matrix cores: ivlm_gemv1_bf12m; copies made on the first decode step, +10 GB for
        # 7B): 2.69 -> 2.34 ms per token.  Matrices whose shape the fragment layout does not take (rows % 16, columns % 64) stay
        # on the bf16 kernel (the lm_head's rows are padded with zeros).  IVLM_DECODE_PACKED=0 / decode_packed = False: bf16 weights everywhere.
        self._decode_packed = os.environ.get("IVLM_DECODE_PACKED", "1") != "0"
        self.decode_attn_parts = os.environ.get("IVLM_DECODE_ATTN_PARTS", "1") != "0"  # (with decode_packed: see _decode_step)
        self.decode_packed_batch = os.environ.get("IVLM_DECODE_PACKED_BATCH", "1") != "0"  # (the batched step: ivlm_gemv16_bf12m)

    # captured decode graphs are kept per (precision, packed decode weights or not): their launches bake in the weight pointers
    def _gkey(self):
        return (self.precision, bool(self._decode_packed))

    @property
    def decode_packed(self):
        return self._decode_packed

    @decode_packed.setter
    def decode_packed(self, flag):
        flag = bool(flag)
        if flag == self._decode_packed:
            return
        self._dgraphs[self._gkey()] = self._dgraph
        self._decode_packed = flag
        self._dgraph = self._dgraphs.get(self._gkey())
        if hasattr(self, "_bgraphs"):
            self._bgraphs = {}

    # ---- which copies of the layer matrices are resident -----------------------------------------
    # Three forms exist: the checkpoint's bf16 matrices (prefill of the bf16 / parity modes, bf16 decode), IEEE fp16 copies (prefill
    # of the fp16-operand default mode), lossless 12-bit planes (the decode step of every mode).  ``prepare`` builds, NOW, what the
    # active mode reads - so that an out-of-memory or an out-of-fp16-range weight surfaces at load / mode switch and not inside the
    # first forward; ``release_unused`` drops what it does not read: in the default mode the bf16 originals of every matrix that
    # has planes (13.5 GB for 7B; `_LayerW` rebuilds one bit for bit from its planes if a non-default path asks for it), in the
    # other modes the fp16 copies.  InteractVLMForCausalLM.set_precision calls both.
    def prepare(self):
        if self.fp8 or self._calibrating:
            return
        for L in self.layers:
            if self._decode_packed:
                for n in _LayerW.MATS:
                    if n + "_p" not in L:
                        L[n + "_p"] = ops.PackedBf12(L[n]) if ops.PackedBf12.takes(*L[n].shape) else None
            if self.precision == "f16":
                self._f16w(L)
            else:  # bf16 / parity prefill reads the bf16 matrices (rebuilt from the planes if the default mode had released them)
                for n in _LayerW.MATS:
                    L[n]
        if self._decode_packed and ops.PackedBf12.takes(16, self.cfg.hidden) and getattr(self, "lm_head_p", None) is None:
            self.lm_head_p = ops.PackedBf12(self.lm_head, pad_rows=True)

    def release_unused(self):
        if self.fp8 or self._calibrating:
            return
        dropped = False
        for L in self.layers:
            for n in _LayerW.MATS:
                if self.precision == "f16":
                    wp = L.get(n + "_p")
                    have16 = n + "_h" in L or n + "_hp" in L
                    if self._decode_packed and wp is not None and wp.frag and dict.__contains__(L, n) and have16:
                        del L[n]
                        dropped = True
                    if n + "_hp" in L and L.pop(n + "_h", None) is not None:  # (the panels replace the row-major copy)
                        dropped = True
                else:
                    for k in (n + "_h", n + "_hp"):
                        if L.pop(k, None) is not None:
                            dropped = True
        if dropped:  # graphs of the bf16-weight decode step hold pointers into what was just released
            self._dgraphs = {k: g for k, g in self._dgraphs.items() if k[1]}
            if not self._decode_packed:
                self._dgraph = None
            if hasattr(self, "_bgraphs"):
                self._bgraphs = {}

    def resident_bytes(self):
        """{form: bytes} of the language model's weights currently in HBM."""
        out = {"bf16": self.embed.numel() * 2 + self.lm_head.numel() * 2 + self.norm.numel() * 2, "f16": 0, "bf12": 0}
        for L in self.layers:
            out["bf16"] += (L["ln1"].numel() + L["ln2"].numel()) * 2
            for n in _LayerW.MATS:
                if dict.__contains__(L, n):
                    out["bf16"] += L[n].numel() * 2
                for k in (n + "_h", n + "_hp"):
                    if L.get(k) is not None:
                        out["f16"] += L[k].numel() * 2
                if L.get(n + "_p") is not None:
                    out["bf12"] += L[n + "_p"].bytes()
        if getattr(self, "lm_head_p", None) is not None:
            out["bf12"] += self.lm_head_p.bytes()
        return out

    # ---- "parity" precision (opt-in): the prefill GEMMs take hi + lo bf16 activation operands, the attention three MFMAs per
    # fragment, and K / V are cached as hi + lo planes (also read by the decode kernels) - no activation is rounded to bf16.
    # "f16": the prefill GEMMs and attention take IEEE fp16 operands (one MFMA pass, an eighth of the bf16 operand rounding; fp16
    # copies of the layer weights, +12.6 GB for 7B, made on first use), K / V are cached as fp16 (the same cache storage, seen
    # as fp16) and read as fp16 by the decode kernels, whose activations are fp32 in every mode.
    def set_precision(self, mode):
        assert mode in ("default", "parity", "f16")
        if mode == self.precision:
            return
        self._dgraphs[self._gkey()] = self._dgraph
        self.precision = mode
        self._dgraph = self._dgraphs.get(self._gkey())
        if hasattr(self, "_bgraphs"):
            self._bgraphs = {}
        if mode == "parity" and self.kcache_lo is None:
            self.kcache_lo, self.vcache_lo = torch.zeros_like(self.kcache), torch.zeros_like(self.vcache)

    # ---- fp8 (OCP e4m3; BASELINE.json configs[4], opt-in) -------------------------------------------------------------------
    # Prefill: e4m3 operands for the four GEMMs of a layer on the MX matrix instruction (per-tensor scales: weights quantised
    # once, activation ranges calibrated on one bf16 pass over a calibration prompt, then fixed; RMSNorm / SwiGLU epilogue write
    # e4m3 directly).  Decode (batch 1): e4m3 WEIGHTS, fp32 activations - the step is HBM-bound on the weight bytes, so this
    # halves its traffic (the products are exact; the only error is the weight quantisation).  lm_head, attention and the KV
    # cache stay bf16; the batched decode (evaluate_batch) keeps bf16 weights.
    fp8 = False
    _calibrating = False

    def enable_fp8(self, calib_x):
        """calib_x fp32 [T, hidden]: input embeddings of a calibration prompt (with its image features spliced in)."""
        dev = self.device
        for L in self.layers:
            for n in ("qkv", "o", "gu", "down"):
                L[n + "_q"], L[n + "_s"] = ops.quantize_fp8(L[n])
            L["amax"] = {k: torch.zeros(1, dtype=F32, device=dev) for k in ("y1", "a", "y2", "h")}
        self._calibrating = True
        try:
            self.forward(calib_x, 0)
        finally:
            self._calibrating = False
        for L in self.layers:
            L["s"] = {k: (v / 448.0).clamp_(min=1e-12) for k, v in L["amax"].items()}
        self.fp8 = True
        self._dgraph = None
        self._dgraphs = {}

    def disable_fp8(self):
        self.fp8 = False
        self._dgraph = None
        self._dgraphs = {}

    def _lo(self, li):
        return (self.kcache_lo[li], self.vcache_lo[li]) if self.precision == "parity" else None

    def _caches(self):
        """this instance's single-sequence cache in the element type of the precision mode (fp16 mode: the same bytes as fp16)"""
        if self.precision == "f16":
            return self.kcache.view(torch.float16), self.vcache.view(torch.float16)
        return self.kcache, self.vcache

    def _f16(self, L):
        """row-major fp16 copies ``*_h`` of the four matrices (the C sequencers' layout; rebuilt from bf16 if they were released)"""
        for n in _LayerW.MATS:
            if n + "_h" not in L:
                L[n + "_h"] = ops.f16_weight(L[n], "llama " + n)
        return L

    # K-panel layout of the fp16 prefill copies: [K / 64, N, 64] instead of [N, K] - the same bytes, a wave's
    # DMA instruction reads 1 KB contiguous.  The prefill GEMMs (M = 330) are bound by each CU's L1 fill path
    # (profiles/r05_gemm_ceilings.txt), which serves contiguous kilobytes faster than eight lines a row stride apart.
    prefill_panels = os.environ.get("IVLM_PREFILL_PANELS", "1") != "0"

    def _f16w(self, L):
        """-> {n: the fp16 weight the Python prefill passes to ops.linear} (panels when ``prefill_panels``, else row-major)"""
        if not self.prefill_panels or self.cfg.hidden % 64 or self.cfg.inter % 64:
            self._f16(L)
            return {n: L[n + "_h"] for n in _LayerW.MATS}
        for n in _LayerW.MATS:
            if n + "_hp" not in L:
                h = L.pop(n + "_h") if n + "_h" in L else ops.f16_weight(L[n], "llama " + n)
                L[n + "_hp"] = ops.panel_weight(h)  # (the row-major copy is not kept: _f16 rebuilds it for the C sequencers)
        return {n: L[n + "_hp"] for n in _LayerW.MATS}

    def _layer_f16(self, L, x, T, pos0, kc, vc, a_out=None):
        """one prefill layer on fp16 operands: x fp32 [T, hidden] -> fp32 [T, hidden]; kc / vc = this layer's fp16 cache planes"""
        c = self.cfg
        H, hd = c.heads, c.hidden // c.heads
        W = self._f16w(L)
        qkv = ops.linear(ops.rmsnorm(x, L["ln1"], c.eps, out_f16=True), W["qkv"], out_f16=True)
        ops.rope_kv(qkv, H, hd, pos0, c.theta, kc, vc, table=self.rope)
        q = qkv.view(T, 3, H, hd)[:, 0].permute(1, 0, 2).unsqueeze(0)
        k = kc[: pos0 + T].permute(1, 0, 2).unsqueeze(0)
        v = vc[: pos0 + T].permute(1, 0, 2).unsqueeze(0)
        a = ops.attention(q, k, v, hd ** -0.5, causal=True, q_pos0=pos0).permute(0, 2, 1, 3).reshape(T, c.hidden)
        x = ops.linear(a, W["o"], residual=x, out_f32=True)
        h = ops.linear(ops.rmsnorm(x, L["ln2"], c.eps, out_f16=True), W["gu"], act="swiglu", out_f16=True)
        return ops.linear(h, W["down"], residual=x, out_f32=True)

    def batch_cache_lo(self, B):
        bc = getattr(self, "_bcache_lo", None)
        if bc is None or bc[0].shape[1] < B:
            kc, vc = self.batch_cache(B)
            bc = self._bcache_lo = (torch.zeros_like(self._bcache[0]), torch.zeros_like(self._bcache[1]))
        return bc[0][:, :B], bc[1][:, :B]

    def _layer_parity(self, L, x, T, pos0, kc, vc, kcl, vcl):
        """one prefill layer on fp32-activation arithmetic: x fp32 [T, hidden] -> fp32 [T, hidden]; kc.. = this layer's cache planes"""
        c = self.cfg
        H, hd = c.heads, c.hidden // c.heads
        qkv = ops.linear(ops.rmsnorm(x, L["ln1"], c.eps, out_split=True), L["qkv"], a_split=True, out_split=True)  # [T, 6*hidden]
        ops.rope_kv_split(qkv, H, hd, pos0, (kc, kcl, vc, vcl), self.rope)
        q6 = qkv.view(T, 2, 3, H, hd)
        qh, ql = (q6[:, i, 0].permute(1, 0, 2).unsqueeze(0) for i in range(2))
        kv = [t[: pos0 + T].permute(1, 0, 2).unsqueeze(0) for t in (kc, kcl, vc, vcl)]
        a = ops.attention_split(qh, ql, kv[0], kv[1], kv[2], kv[3], hd ** -0.5, causal=True, q_pos0=pos0)  # [T, 2*hidden]
        x = ops.linear(a, L["o"], residual=x, out_f32=True, a_split=True)
        h = ops.linear(ops.rmsnorm(x, L["ln2"], c.eps, out_split=True), L["gu"], act="swiglu", a_split=True, out_split=True)
        return ops.linear(h, L["down"], residual=x, out_f32=True, a_split=True)

    def embed_ids(self, ids_i32, out=None):
        """embed_tokens gather: ids int32 [n] -> fp32 [n, hidden] (the start of the fp32 residual stream)."""
        return ops.gather_rows(self.embed, ids_i32, out=out, out_kind="f32")

    def forward(self, x, pos0, cache=None):
        """x fp32 [T, hidden] input embeddings at positions pos0..pos0+T-1 -> final-norm hidden fp32 [T, hidden];
        appends to the KV cache (prefill: T = prompt, decode: T = 1).  cache = (k, v) [layers, Tmax, H, hd] views of
        another sequence's slab (batched generation); default: this instance's single-sequence cache.
        Precision: the residual stream is fp32 (GEMM residual epilogues write fp32, RMSNorm reads it); the MFMA operands
        (normed rows, attention output, SwiGLU product) are bf16."""
        c = self.cfg
        T = x.shape[0]
        H, hd = c.heads, c.hidden // c.heads
        assert pos0 + T <= self.max_len and x.dtype == F32
        if T == 1 and cache is None:
            return self._decode_step(x, pos0)
        if 1 < T <= 16 and cache is None and self.precision != "default":
            # (the fp16 / split tile GEMMs need M > 16: a short chunk goes token by token through the fp32-activation decode
            #  kernels, which are exact on the bf16 weights)
            return torch.cat([self._decode_step(x[t: t + 1], pos0 + t) for t in range(T)], 0)
        kc, vc = cache[:2] if cache is not None else self._caches()
        if self.precision == "f16":
            for li, L in enumerate(self.layers):
                x = self._layer_f16(L, x, T, pos0, kc[li], vc[li])
            return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
        if self.precision == "parity":
            kcl, vcl = cache[2:] if cache is not None else (self.kcache_lo, self.vcache_lo)
            for li, L in enumerate(self.layers):
                x = self._layer_parity(L, x, T, pos0, kc[li], vc[li], kcl[li], vcl[li])
            return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
        fp8 = self.fp8 and T > 16 and not self._calibrating
        for li, L in enumerate(self.layers):
            cal = L.get("amax") if self._calibrating else None
            sc = L["s"] if fp8 else None
            if fp8:
                qkv = ops.linear_fp8(ops.rmsnorm(x, L["ln1"], c.eps, fp8_scale=sc["y1"]), L["qkv_q"], sc["y1"], L["qkv_s"])
            else:
                y = ops.rmsnorm(x, L["ln1"], c.eps)
                qkv = ops.linear(y, L["qkv"])  # [T, 3*hidden] == [T, 3, H, hd]
            ops.rope_kv(qkv, H, hd, pos0, c.theta, kc[li], vc[li], table=self.rope)
            q = qkv.view(T, 3, H, hd)[:, 0].permute(1, 0, 2).unsqueeze(0)  # [1,H,T,hd]
            k = kc[li, : pos0 + T].permute(1, 0, 2).unsqueeze(0)
            v = vc[li, : pos0 + T].permute(1, 0, 2).unsqueeze(0)
            a = ops.attention(q, k, v, hd ** -0.5, causal=True, q_pos0=pos0).permute(0, 2, 1, 3).reshape(T, c.hidden)
            if fp8:
                x = ops.linear_fp8(ops.gather_rows(a, out_kind="fp8", scale=sc["a"]), L["o_q"], sc["a"], L["o_s"], residual=x,
                                   out_kind="f32")
                h8 = ops.linear_fp8(ops.rmsnorm(x, L["ln2"], c.eps, fp8_scale=sc["y2"]), L["gu_q"], sc["y2"], L["gu_s"],
                                    act="swiglu", out_kind="fp8", scale_out=sc["h"])
                x = ops.linear_fp8(h8, L["down_q"], sc["h"], L["down_s"], residual=x, out_kind="f32")
                continue
            x = ops.linear(a, L["o"], residual=x, out_f32=True)
            y2 = ops.rmsnorm(x, L["ln2"], c.eps)
            h = ops.linear(y2, L["gu"], act="swiglu")
            if cal:
                for k_, t_ in (("y1", y), ("a", a), ("y2", y2), ("h", h)):
                    ops.amax(t_, cal[k_])
            x = ops.linear(h, L["down"], residual=x, out_f32=True)
        return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)

    def forward_packed(self, xs, kc, vc, lo=None):
        """Prefill of B sequences in ONE pass over the weights: xs = [x_b fp32 [T_b, hidden]] (lengths may differ), kc / vc
        [layers, B, Tmax, H, hd] cache slabs -> [final-norm hidden fp32 [T_b, hidden]].  The four projections of a layer run
        on the packed rows (M = sum T_b: one efficient GEMM instead of B skinny ones, weights streamed once); RoPE + cache
        append and the causal attention stay per sequence.  Row-wise identical arithmetic to ``forward``."""
        c = self.cfg
        H, hd = c.heads, c.hidden // c.heads
        lens = [int(x.shape[0]) for x in xs]
        assert max(lens) <= self.max_len
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        x = torch.cat(xs, 0)
        if self.precision == "f16":  # fp16 operands (see _layer_f16); kc / vc are fp16 views of the slabs
            for li, L in enumerate(self.layers):
                W = self._f16w(L)
                qkv = ops.linear(ops.rmsnorm(x, L["ln1"], c.eps, out_f16=True), W["qkv"], out_f16=True)
                a = torch.empty(offs[-1], c.hidden, dtype=torch.float16, device=x.device)
                for b, T in enumerate(lens):
                    qb = qkv[offs[b]: offs[b + 1]]
                    ops.rope_kv(qb, H, hd, 0, c.theta, kc[li, b], vc[li, b], table=self.rope)
                    q = qb.view(T, 3, H, hd)[:, 0].permute(1, 0, 2).unsqueeze(0)
                    k = kc[li, b, :T].permute(1, 0, 2).unsqueeze(0)
                    v = vc[li, b, :T].permute(1, 0, 2).unsqueeze(0)
                    ops.attention(q, k, v, hd ** -0.5, causal=True, q_pos0=0,
                                  out=a[offs[b]: offs[b + 1]].view(1, T, H, hd).permute(0, 2, 1, 3))
                x = ops.linear(a, W["o"], residual=x, out_f32=True)
                h = ops.linear(ops.rmsnorm(x, L["ln2"], c.eps, out_f16=True), W["gu"], act="swiglu", out_f16=True)
                x = ops.linear(h, W["down"], residual=x, out_f32=True)
            x = ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
            return [x[offs[b]: offs[b + 1]] for b in range(len(lens))]
        if self.precision == "parity":  # fp32-activation arithmetic (see _layer_parity), lo = (kc_lo, vc_lo) slabs
            kcl, vcl = lo
            for li, L in enumerate(self.layers):
                qkv = ops.linear(ops.rmsnorm(x, L["ln1"], c.eps, out_split=True), L["qkv"], a_split=True, out_split=True)
                a = torch.empty(offs[-1], 2 * c.hidden, dtype=BF16, device=x.device)
                for b, T in enumerate(lens):
                    qb = qkv[offs[b]: offs[b + 1]]
                    ops.rope_kv_split(qb, H, hd, 0, (kc[li, b], kcl[li, b], vc[li, b], vcl[li, b]), self.rope)
                    q6 = qb.view(T, 2, 3, H, hd)
                    qh, ql = (q6[:, i, 0].permute(1, 0, 2).unsqueeze(0) for i in range(2))
                    kv = [t[li, b, :T].permute(1, 0, 2).unsqueeze(0) for t in (kc, kcl, vc, vcl)]
                    ops.attention_split(qh, ql, kv[0], kv[1], kv[2], kv[3], hd ** -0.5, causal=True, q_pos0=0,
                                        out=a[offs[b]: offs[b + 1]].view(1, T, 2, H, hd))
                x = ops.linear(a, L["o"], residual=x, out_f32=True, a_split=True)
                h = ops.linear(ops.rmsnorm(x, L["ln2"], c.eps, out_split=True), L["gu"], act="swiglu", a_split=True, out_split=True)
                x = ops.linear(h, L["down"], residual=x, out_f32=True, a_split=True)
            x = ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
            return [x[offs[b]: offs[b + 1]] for b in range(len(lens))]
        for li, L in enumerate(self.layers):
            qkv = ops.linear(ops.rmsnorm(x, L["ln1"], c.eps), L["qkv"])
            a = torch.empty(offs[-1], c.hidden, dtype=BF16, device=x.device)
            for b, T in enumerate(lens):
                qb = qkv[offs[b]: offs[b + 1]]
                ops.rope_kv(qb, H, hd, 0, c.theta, kc[li, b], vc[li, b], table=self.rope)
                q = qb.view(T, 3, H, hd)[:, 0].permute(1, 0, 2).unsqueeze(0)
                k = kc[li, b, :T].permute(1, 0, 2).unsqueeze(0)
                v = vc[li, b, :T].permute(1, 0, 2).unsqueeze(0)
                ops.attention(q, k, v, hd ** -0.5, causal=True, q_pos0=0,
                              out=a[offs[b]: offs[b + 1]].view(1, T, H, hd).permute(0, 2, 1, 3))
            x = ops.linear(a, L["o"], residual=x, out_f32=True)
            h = ops.linear(ops.rmsnorm(x, L["ln2"], c.eps), L["gu"], act="swiglu")
            x = ops.linear(h, L["down"], residual=x, out_f32=True)
        x = ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
        return [x[offs[b]: offs[b + 1]] for b in range(len(lens))]

    # ---- one decode step as a replayable HIP graph ------------------------------------------------------------------
    # Static buffers: token id in, position (device int32, read by the attention kernel), hidden out, argmax out.
    # Per generated token the host then issues 1 graph launch instead of ~165 kernel launches (2.4 ms of Python).
    def decode_graph(self):
        if self._dgraph is None:
            dev = self.device
            st = dict(tok=torch.zeros(1, dtype=torch.int32, device=dev), pos=torch.zeros(1, dtype=torch.int32, device=dev),
                      pos64=torch.zeros(1, dtype=torch.int64, device=dev))
            c = self.cfg
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            # (every block of the fused launch must be resident at once: one 1024-thread block per CU)
            if self.fuse_attn_oproj and c.hidden in (512, 1024, 4096, 5120) and c.heads + c.hidden // 32 <= n_cu:
                # fused attention + o_proj launches: per-layer arrival counters, tokens-decoded counter, status word
                st["fused"] = dict(step=torch.zeros(1, dtype=torch.int32, device=dev),
                                   counters=torch.zeros(c.layers, 32, dtype=torch.int32, device=dev),  # 128-B apart
                                   status=torch.zeros(1, dtype=torch.int32, device=dev),
                                   scratch=torch.zeros(c.layers, c.hidden, dtype=F32, device=dev))
            self._fused = st.get("fused")

            def body():
                e = self.embed_ids(st["tok"])
                h = self._decode_step(e, st["pos"])
                st["hidden"] = h
                st["nxt"] = ops.argmax(self.logits(h), bump=st["pos"])  # (+ position += 1 in the same launch)
                if self._fused is not None:
                    self._fused["step"].add_(1)

            caches = [self.kcache, self.vcache] + ([self.kcache_lo, self.vcache_lo] if self.precision == "parity" else [])
            saved = [t[:, :1].clone() for t in caches]  # the warm-up / capture runs write row 0 (fp16 mode: the same bytes)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            st["pos"].zero_()
            st["pos64"].zero_()
            if self._fused is not None:
                self._fused["step"].zero_()
                self._fused["counters"].zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            for t, sv in zip(caches, saved):
                t[:, :1].copy_(sv)
            st["graph"] = g
            self._dgraph = st
        return self._dgraph

    # ---- B sequences per decode step (configs[2]: 8 images per GPU) ---------------------------------------------------
    # The weights are streamed once per step for all B tokens (GEMV rows M = B <= 8 share every weight load), so a step of
    # 8 sequences costs little more than a step of one.  Each sequence owns a cache slab [Tmax, H, hd] per layer.
    def batch_cache(self, B):
        bc = getattr(self, "_bcache", None)
        if bc is None or bc[0].shape[1] < B:
            c = self.cfg
            H, hd = c.heads, c.hidden // c.heads
            bc = self._bcache = tuple(torch.zeros(c.layers, B, self.max_len, H, hd, dtype=BF16, device=self.device)
                                      for _ in range(2))
            self._bgraphs = {}
        if self.precision == "f16":  # the same slabs, seen as fp16
            return bc[0].view(torch.float16)[:, :B], bc[1].view(torch.float16)[:, :B]
        return bc[0][:, :B], bc[1][:, :B]

    def decode_step_batch(self, x, pos_dev, kc, vc, lo=None):
        """x fp32 [B, hidden] (one new token per sequence), pos_dev int32 [B], kc/vc [layers, B, Tmax, H, hd] ->
        final-norm hidden fp32 [B, hidden]; same arithmetic per row as ``_decode_step`` (fp32 activations: the skinny MFMA
        kernel splits them into hi + lo bf16 operands, the batch-1 GEMV multiplies them exactly - equal to ~1e-5)."""
        c = self.cfg
        H, hd = c.heads, c.hidden // c.heads
        if x.shape[0] > 16:
            raise ops.IvlmError("decode_step_batch: at most 16 sequences per step (weight-streaming kernels)")
        packed = self.decode_packed and self.decode_packed_batch and x.shape[0] > 1

        def lin(x_, L, n, **kw):  # (the packed planes of the batch-1 step where the fragment layout takes the matrix, bf16 otherwise)
            if packed:
                if n + "_p" not in L:
                    L[n + "_p"] = ops.PackedBf12(L[n]) if ops.PackedBf12.takes(*L[n].shape) else None
                if L[n + "_p"] is not None and L[n + "_p"].frag:
                    return ops.linear_bf12(x_, L[n + "_p"], **kw)
            return ops.linear(x_, L[n], out_f32=True, **kw)

        for li, L in enumerate(self.layers):
            qkv = lin(x, L, "qkv", rms=(L["ln1"], c.eps))
            a = ops.llama_decode_attn_batch(qkv, kc[li], vc[li], H, hd, pos_dev, c.theta, hd ** -0.5, table=self.rope,
                                            lo=(lo[0][li], lo[1][li]) if lo is not None else None)
            x = lin(a, L, "o", residual=x)
            h = lin(x, L, "gu", act="swiglu", rms=(L["ln2"], c.eps))
            x = lin(h, L, "down", residual=x)
        return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)

    def decode_graph_batch(self, B):
        """One batched decode step (embed -> layers -> norm -> lm_head -> argmax, positions += 1) as a HIP graph."""
        kc, vc = self.batch_cache(B)
        lo = self.batch_cache_lo(B) if self.precision == "parity" else None
        st = self._bgraphs.get(B)
        if st is None:
            dev = self.device
            st = dict(tok=torch.zeros(B, dtype=torch.int32, device=dev), pos=torch.zeros(B, dtype=torch.int32, device=dev))

            def body():
                h = self.decode_step_batch(self.embed_ids(st["tok"]), st["pos"], kc, vc, lo)
                st["hidden"] = h
                st["nxt"] = ops.argmax(self.logits(h), bump=st["pos"])  # (+ positions += 1 in the same launch)

            caches = [kc, vc] + (list(lo) if lo is not None else [])
            saved = [t[:, :, :1].clone() for t in caches]  # the warm-up / capture runs write row 0 of every slab
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            st["pos"].zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            for t, sv in zip(caches, saved):
                t[:, :, :1].copy_(sv)
            st["pos"].zero_()
            st["graph"] = g
            self._bgraphs[B] = st
        return st

    def _attn_scratch(self):
        """Scratch of the split-KV decode attention, or None = the one-block-per-head kernel.  Opt-in (`decode_splitkv = True` before
        the first decode step, or IVLM_DECODE_SPLITKV=1): measured on the 7B shapes it is even at 330 cached positions (the headline
        prompt: 2.72 vs 2.70 ms/token) and ahead from ~450 (640 positions: 2.79 vs 2.91 ms/token)."""
        if not (self.decode_splitkv or os.environ.get("IVLM_DECODE_SPLITKV", "0") == "1") or self.precision == "parity":
            return None
        if getattr(self, "_dec_scratch", None) is None:
            self._dec_scratch = ops.decode_attn_scratch(self.cfg.heads, self.cfg.hidden // self.cfg.heads, self.norm.device)
        return self._dec_scratch

    def _decode_step(self, x, pos):
        """One new token, x fp32 [1, hidden]: 4 launches per layer (RMSNorm fused into the q|k|v and gate|up GEMVs, RoPE +
        cache append + attention + o_proj + residual in one launch, SwiGLU and the other residual add in GEMV epilogues).
        Every activation between the weight-streaming kernels is fp32 and the bf16-weight x fp32-activation products are
        exact: no operand rounding on the decode path (it is HBM-bound; the fp32 traffic is a few hundred KB per token)."""
        c = self.cfg
        H, hd = c.heads, c.hidden // c.heads
        fz = self._fused if isinstance(pos, torch.Tensor) else None
        if self.fp8:  # e4m3 weights, fp32 activations: half the bytes per token
            for li, L in enumerate(self.layers):
                qkv = ops.linear_fp8w(x, L["qkv_q"], L["qkv_s"], rms=(L["ln1"], c.eps))
                a = ops.llama_decode_attn(qkv, self.kcache[li], self.vcache[li], H, hd, pos, c.theta, hd ** -0.5, table=self.rope)
                x = ops.linear_fp8w(a, L["o_q"], L["o_s"], residual=x)
                h = ops.linear_fp8w(x, L["gu_q"], L["gu_s"], act="swiglu", rms=(L["ln2"], c.eps))
                x = ops.linear_fp8w(h, L["down_q"], L["down_s"], residual=x)
            return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
        kc_, vc_ = self._caches()
        if self._decode_packed and fz is None:
            def lin(x_, L, n, **kw):  # (packed where the fragment layout takes the matrix, bf16 otherwise)
                if n + "_p" not in L:
                    N_, K_ = L[n].shape
                    L[n + "_p"] = ops.PackedBf12(L[n]) if ops.PackedBf12.takes(N_, K_) else None
                wp = L[n + "_p"]
                return ops.linear_bf12(x_, wp, **kw) if wp is not None else ops.linear(x_, L[n], out_f32=True, **kw)

            # attention as 4 key ranges per head whose (o, max, sum) partials the o_proj merges in its prologue (128 blocks instead of
            # 32, no merge launch): when o_proj is packed and K / V are single planes (not the hi + lo planes of "parity")
            split = self.decode_attn_parts and self._lo(0) is None
            if split and getattr(self, "_parts", None) is None:
                self._parts = torch.zeros(H * 4 * (hd + 4), dtype=F32, device=self.norm.device)
            for li, L in enumerate(self.layers):
                qkv = lin(x, L, "qkv", rms=(L["ln1"], c.eps))
                if "o_p" not in L:
                    L["o_p"] = ops.PackedBf12(L["o"]) if ops.PackedBf12.takes(*L["o"].shape) else None
                if split and L["o_p"] is not None:
                    ops.llama_decode_attn_parts(qkv, kc_[li], vc_[li], H, hd, pos, c.theta, hd ** -0.5, self._parts, table=self.rope)
                    x = ops.linear_bf12(None, L["o_p"], residual=x, parts=(self._parts, hd))
                else:
                    a = ops.llama_decode_attn(qkv, kc_[li], vc_[li], H, hd, pos, c.theta, hd ** -0.5, table=self.rope,
                                              lo=self._lo(li), scratch=self._attn_scratch())
                    x = lin(a, L, "o", residual=x)
                h = lin(x, L, "gu", act="swiglu", rms=(L["ln2"], c.eps))
                x = lin(h, L, "down", residual=x)
            return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)
        for li, L in enumerate(self.layers):
            qkv = ops.linear(x, L["qkv"], rms=(L["ln1"], c.eps), out_f32=True)
            if fz is not None and self.precision == "default":  # attention + o_proj + residual in one launch (W_o streams while the attention runs)
                x = ops.llama_attn_oproj(qkv, self.kcache[li], self.vcache[li], L["o"], x, H, hd, pos, fz["step"],
                                         fz["counters"][li], fz["status"], c.theta, hd ** -0.5, self.rope, fz["scratch"][li])
            else:
                a = ops.llama_decode_attn(qkv, kc_[li], vc_[li], H, hd, pos, c.theta, hd ** -0.5, table=self.rope, lo=self._lo(li),
                                          scratch=self._attn_scratch())
                x = ops.linear(a, L["o"], residual=x, out_f32=True)
            h = ops.linear(x, L["gu"], act="swiglu", rms=(L["ln2"], c.eps), out_f32=True)
            x = ops.linear(h, L["down"], residual=x, out_f32=True)
        return ops.rmsnorm(x, self.norm, c.eps, out_f32=True)

    def logits(self, hidden_rows):
        """lm_head on fp32 [n, hidden] -> f32 [n, vocab].  n <= 16: exact fp32 activations on the weight-streaming kernels; more rows:
        one tile GEMM on [hi | lo] bf16 operands (fp32-equivalent activations, the weight matrix read once - ADVICE r3: 16-row
        chunks re-streamed the 262 MB of lm_head once per chunk)."""
        if hidden_rows.shape[0] > 16:
            return ops.linear(ops.split_rows(hidden_rows.contiguous()), self.lm_head, out_f32=True, a_split=True)
        if hidden_rows.shape[0] == 1 and self.decode_packed and ops.PackedBf12.takes(16, self.cfg.hidden):
            if getattr(self, "lm_head_p", None) is None:  # (the decode step's lm_head: 12-bit weights, rows padded to 16 with zeros)
                self.lm_head_p = ops.PackedBf12(self.lm_head, pad_rows=True)
            return ops.linear_bf12(hidden_rows.contiguous(), self.lm_head_p)
        return ops.linear(hidden_rows, self.lm_head, out_f32=True)
