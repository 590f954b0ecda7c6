"""ctypes mirror of the stage-level C entry points (``ivlm_llama_prefill`` / ``ivlm_llama_decode_step``, include/ivlm_hip.h):
what a non-Python caller binds.  ``interactvlm_amd.llava.Llama`` sequences the same kernels from Python; this module exists to
exercise the C sequencers (tests/test_stages_gpu.py) and as the binding example of INTEGRATION.md."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


class LlamaCfg(C.Structure):
    _fields_ = [("layers", C.c_int), ("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("max_len", C.c_int),
                ("eps", C.c_float), ("theta", C.c_float), ("fuse_attn_oproj", C.c_int)]


class LlamaLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1", "qkv", "o", "ln2", "gu", "down")]


class Bf12M(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Pf", "Ef", "ebase", "patch_ptr", "patch_col", "patch_val")]


class LlamaLayerBf12(C.Structure):
    _fields_ = [("ln1", C.c_void_p), ("ln2", C.c_void_p), ("qkv", Bf12M), ("o", Bf12M), ("gu", Bf12M), ("down", Bf12M)]


class LlamaStages:
    """Wraps a ``llava.Llama`` instance's weights / caches / rope tables as the C structs and calls the C sequencers."""

    def __init__(self, llm):
        self.llm = llm
        c = llm.cfg
        self.cfg = LlamaCfg(c.layers, c.hidden, c.heads, c.inter, llm.max_len, c.eps, c.theta, 1 if llm.fuse_attn_oproj else 0)
        # (the structs hold raw device pointers: keep the tensors alive - the host model may release weight copies its active
        #  precision mode does not read, llava.Llama.release_unused)
        self._keep = [[L[k] for k in ("ln1", "qkv", "o", "ln2", "gu", "down")] for L in llm.layers]
        self.layers = (LlamaLayer * c.layers)(*[LlamaLayer(*[t.data_ptr() for t in ts]) for ts in self._keep])
        lib = _lib.load()
        self._dws = torch.zeros(lib.ivlm_llama_decode_workspace_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=llm.device)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def prefill(self, x, pos0=0):
        """x fp32 [T, hidden] -> final-norm hidden fp32 [T, hidden]; appends to the instance's KV cache."""
        lib = _lib.load()
        llm = self.llm
        T = x.shape[0]
        x = x.contiguous()
        out = torch.empty_like(x)
        nbytes = lib.ivlm_llama_prefill_workspace_bytes(C.byref(self.cfg), T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib.ivlm_llama_prefill(C.byref(self.cfg), self.layers, llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                     llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(), T,
                                     int(pos0), out.data_ptr(), ws.data_ptr(), nbytes, self._stream()), "llama_prefill")
        return out

    def _layers16(self):
        """the layer table with qkv / o / gu / down pointing to the fp16 copies of the weights (ivlm_llama_prefill_f16)"""
        if not hasattr(self, "layers16"):
            Ls = [self.llm._f16(L) for L in self.llm.layers]
            self._keep16 = [[L[n + "_h"] for n in ("qkv", "o", "gu", "down")] for L in Ls]
            self.layers16 = (LlamaLayer * len(Ls))(*[LlamaLayer(L["ln1"].data_ptr(), L["qkv_h"].data_ptr(), L["o_h"].data_ptr(),
                                                                 L["ln2"].data_ptr(), L["gu_h"].data_ptr(), L["down_h"].data_ptr())
                                                      for L in Ls])
        return self.layers16

    def prefill_f16(self, x, pos0=0):
        """The default precision of the host model: fp16 MFMA operands, fp16 KV cache (the instance's cache storage seen as fp16)."""
        lib = _lib.load()
        llm = self.llm
        T = x.shape[0]
        x = x.contiguous()
        out = torch.empty_like(x)
        nbytes = lib.ivlm_llama_prefill_workspace_bytes(C.byref(self.cfg), T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib.ivlm_llama_prefill_f16(C.byref(self.cfg), self._layers16(), llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                         llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(), T,
                                         int(pos0), out.data_ptr(), ws.data_ptr(), nbytes, self._stream()), "llama_prefill_f16")
        return out

    def decode_step_f16kv(self, x, pos_dev, advance=True):
        """One decode step against the fp16 KV cache (bf16 weights, fp32 activations)."""
        lib = _lib.load()
        llm = self.llm
        out = torch.empty_like(x)
        check(lib.ivlm_llama_decode_step_f16kv(C.byref(self.cfg), self.layers, llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                               llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(),
                                               pos_dev.data_ptr(), 1 if advance else 0, out.data_ptr(), self._dws.data_ptr(),
                                               self._dws.numel(), self._stream()), "llama_decode_step_f16kv")
        return out

    def _layers_bf12(self):
        """the layer table of the packed decode step: every matrix as ops.PackedBf12 in the fragment layout (packed here if the
        host model has not decoded yet)"""
        if not hasattr(self, "layers_bf12"):
            from . import ops

            rows = []
            self._keep12 = []
            for L in self.llm.layers:
                for n in ("qkv", "o", "gu", "down"):
                    if L.get(n + "_p") is None:
                        L[n + "_p"] = ops.PackedBf12(L[n])
                    self._keep12.append(L[n + "_p"])
                    if not L[n + "_p"].frag:
                        raise ops.IvlmError("decode_step_bf12: a matrix does not take the fragment layout (rows % 16, columns % 64)")
                rows.append(LlamaLayerBf12(L["ln1"].data_ptr(), L["ln2"].data_ptr(),
                                           *[Bf12M(*L[n + "_p"]._args_frag()) for n in ("qkv", "o", "gu", "down")]))
            self.layers_bf12 = (LlamaLayerBf12 * len(rows))(*rows)
        return self.layers_bf12

    def decode_step_bf12(self, x, pos_dev, advance=True):
        """One decode step on losslessly packed weights (the host model's default decode path) against the instance's KV cache in the
        element type of its precision mode (fp16 after prefill_f16, bf16 after prefill)."""
        lib = _lib.load()
        llm = self.llm
        out = torch.empty_like(x)
        cache_dt = 4 if llm.precision == "f16" else 1  # IVLM_F16 | IVLM_BF16
        check(lib.ivlm_llama_decode_step_bf12(C.byref(self.cfg), self._layers_bf12(), llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                              llm.vcache.data_ptr(), cache_dt, llm.rope[0].data_ptr(), llm.rope[1].data_ptr(),
                                              x.data_ptr(), pos_dev.data_ptr(), 1 if advance else 0, out.data_ptr(),
                                              self._dws.data_ptr(), self._dws.numel(), self._stream()), "llama_decode_step_bf12")
        return out

    def start_generation(self):
        self._dws.zero_()

    def decode_step(self, x, pos_dev, advance=True):
        """x fp32 [1, hidden], pos_dev int32 [1] on the device -> hidden fp32 [1, hidden]."""
        lib = _lib.load()
        llm = self.llm
        out = torch.empty_like(x)
        check(lib.ivlm_llama_decode_step(C.byref(self.cfg), self.layers, llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                         llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(),
                                         pos_dev.data_ptr(), 1 if advance else 0, out.data_ptr(), self._dws.data_ptr(),
                                         self._dws.numel(), self._stream()), "llama_decode_step")
        return out


class ClipCfgC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("layers_run", "hidden", "heads", "inter", "image_size", "patch", "kpad", "tokens")] + [
        ("eps", C.c_float)]


class ClipHeadC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("patch_w", "pos", "cls_row", "pre_ln_w", "pre_ln_b")]


class ClipLayerC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b")]


class ClipStages:
    """``ivlm_clip_encode`` over the weights of a ``llava.ClipTower``."""

    def __init__(self, tower):
        self.t = tower
        c = tower.cfg
        self.cfg = ClipCfgC(len(tower.layers), c.hidden, c.heads, c.inter, c.image_size, c.patch, tower.kpad, c.tokens, c.eps)
        p = lambda t: t.data_ptr()
        self.head = ClipHeadC(p(tower.patch_w), p(tower.pos), p(tower.cls_row), p(tower.pre_ln.w), p(tower.pre_ln.b))
        self.layers = (ClipLayerC * len(tower.layers))(*[
            ClipLayerC(p(L["ln1"].w), p(L["ln1"].b), p(L["qkv_w"]), p(L["qkv_b"]), p(L["out"].w), p(L["out"].b), p(L["ln2"].w),
                       p(L["ln2"].b), p(L["fc1"].w), p(L["fc1"].b), p(L["fc2"].w), p(L["fc2"].b)) for L in tower.layers])

    def __call__(self, images, precision="bf16"):
        """precision 'f16' (the host model's default): ``ivlm_clip_encode_f16`` -> [B, tokens-1, 2*hidden] split rows."""
        lib = _lib.load()
        images = images.to(torch.bfloat16).contiguous()
        B = images.shape[0]
        c = self.t.cfg
        f16 = precision == "f16"
        out = torch.empty(B, c.tokens - 1, (2 if f16 else 1) * c.hidden, dtype=torch.bfloat16, device=images.device)
        nbytes = lib.ivlm_clip_encode_workspace_bytes(C.byref(self.cfg), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device)
        if f16:
            if not hasattr(self, "layers16"):
                p = lambda t: t.data_ptr()
                Ls = [self.t._f16(L) for L in self.t.layers]
                self.layers16 = (ClipLayerC * len(Ls))(*[
                    ClipLayerC(p(L["ln1"].w), p(L["ln1"].b), p(L["qkv_h"]), p(L["qkv_b"]), p(L["out_h"]), p(L["out"].b), p(L["ln2"].w),
                               p(L["ln2"].b), p(L["fc1_h"]), p(L["fc1"].b), p(L["fc2_h"]), p(L["fc2"].b)) for L in Ls])
            check(lib.ivlm_clip_encode_f16(C.byref(self.cfg), C.byref(self.head), self.layers16, images.data_ptr(), B, out.data_ptr(),
                                           ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "clip_encode_f16")
            return out
        check(lib.ivlm_clip_encode(C.byref(self.cfg), C.byref(self.head), self.layers, images.data_ptr(), B, out.data_ptr(),
                                   ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "clip_encode")
        return out


class SamCfgC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("embed_dim", "depth", "heads", "grid", "window", "patch", "img_size", "out_chans", "mlp_dim")]


class SamHeadC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("patch_w", "patch_b", "pos_embed", "neck0_w", "neck1_w", "neck1_b", "neck2_w", "neck3_w",
                                          "neck3_b")]


class SamBlockC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "rel_h", "rel_w", "rel_cat", "proj_w", "proj_b",
                                          "norm2_w", "norm2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b")] + [("global_attn", C.c_int)]


class SamMlpF16C(C.Structure):
    _fields_ = [("lin1_w16", C.c_void_p), ("lin2_w16", C.c_void_p)]


class SamBlockF16C(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w16", "proj_w16", "lin1_w16", "lin2_w16", "qkv_b16", "rel_cat16")]


class SamEncodeStages:
    """``ivlm_sam_encode`` over the weights of a ``sam.SamImageEncoder`` (bf16 operands, fp32 residual stream)."""

    def __init__(self, enc):
        from . import ops

        self.e = enc
        c = enc.cfg
        self.cfg = SamCfgC(c.embed_dim, c.depth, c.num_heads, c.grid, c.window, c.patch, c.img_size, c.out_chans,
                           enc.blocks[0]["lin1"].w.shape[0])
        p = lambda t: t.data_ptr()
        self.head = SamHeadC(p(enc.patch.w), p(enc.patch.b), p(enc.pos_embed), p(enc.neck0.w), p(enc.neck1.w), p(enc.neck1.b),
                             p(enc.neck2_w), p(enc.neck3.w), p(enc.neck3.b))
        for blk in enc.blocks:
            if "rel_cat" not in blk:
                blk["rel_cat"] = ops.relpos_tables_cat(blk["rel_h"], blk["rel_w"])
        self.blocks = (SamBlockC * c.depth)(*[
            SamBlockC(p(b["norm1"].w), p(b["norm1"].b), p(b["qkv"].w), p(b["qkv"].b), p(b["rel_h"]), p(b["rel_w"]), p(b["rel_cat"]),
                      p(b["proj"].w), p(b["proj"].b), p(b["norm2"].w), p(b["norm2"].b), p(b["lin1"].w), p(b["lin1"].b),
                      p(b["lin2"].w), p(b["lin2"].b), 1 if b["glob"] else 0) for b in enc.blocks])

    def __call__(self, images, precision="default"):
        """precision 'parity': ``ivlm_sam_encode_parity`` (fp32-activation arithmetic, see SamImageEncoder.precision);
        'parity-encoder': ``ivlm_sam_encode_parity_f16mlp`` (the same with the MLP GEMMs on fp16 operands)."""
        lib = _lib.load()
        images = images.to(torch.bfloat16).contiguous()
        V = images.shape[0]
        c = self.e.cfg
        out = torch.empty(V, c.grid * c.grid, c.out_chans, dtype=torch.float32, device=images.device)
        st = torch.cuda.current_stream().cuda_stream
        if precision == "f16":  # the host model's default: fp16 operands, exact q path (ivlm_sam_encode_f16)
            from . import ops
            if not hasattr(self, "blocks16"):
                rows = []
                for b in self.e.blocks:
                    w = self.e._f16_weights(b, ("qkv", "proj", "lin1", "lin2"))
                    if "qkv_b_h" not in b:
                        b["qkv_b_h"] = ops.bf16_to_f16(b["qkv"].b)
                    if "rel_cat_h" not in b:
                        b["rel_cat_h"] = ops.bf16_to_f16(b["rel_cat"])
                    rows.append(SamBlockF16C(*[t.data_ptr() for t in w], b["qkv_b_h"].data_ptr(), b["rel_cat_h"].data_ptr()))
                self.blocks16 = (SamBlockF16C * c.depth)(*rows)
            nbytes = lib.ivlm_sam_encode_f16_workspace_bytes(C.byref(self.cfg), V)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device)
            check(lib.ivlm_sam_encode_f16(C.byref(self.cfg), C.byref(self.head), self.blocks, self.blocks16, images.data_ptr(), V,
                                          out.data_ptr(), ws.data_ptr(), nbytes, st), "sam_encode_f16")
            return out
        if precision == "parity-encoder":
            if not hasattr(self, "mlp16"):
                w16 = [self.e._f16_weights(b) for b in self.e.blocks]
                self.mlp16 = (SamMlpF16C * c.depth)(*[SamMlpF16C(a.data_ptr(), b.data_ptr()) for a, b in w16])
            nbytes = lib.ivlm_sam_encode_parity_workspace_bytes(C.byref(self.cfg), V)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device)
            check(lib.ivlm_sam_encode_parity_f16mlp(C.byref(self.cfg), C.byref(self.head), self.blocks, self.mlp16, images.data_ptr(),
                                                    V, out.data_ptr(), ws.data_ptr(), nbytes, st), "sam_encode_parity_f16mlp")
            return out
        size_fn, fn = ((lib.ivlm_sam_encode_parity_workspace_bytes, lib.ivlm_sam_encode_parity) if precision == "parity"
                       else (lib.ivlm_sam_encode_workspace_bytes, lib.ivlm_sam_encode))
        nbytes = size_fn(C.byref(self.cfg), V)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device)
        check(fn(C.byref(self.cfg), C.byref(self.head), self.blocks, images.data_ptr(), V, out.data_ptr(), ws.data_ptr(), nbytes,
                 st), "sam_encode")
        return out


class LinC(C.Structure):
    _fields_ = [("w2", C.c_void_p), ("b", C.c_void_p), ("n", C.c_int), ("k", C.c_int)]


class DecAttnC(C.Structure):
    _fields_ = [(n, LinC) for n in ("q", "k", "v", "o")]


class DecLayerC(C.Structure):
    _fields_ = ([("self_attn", DecAttnC), ("t2i", DecAttnC), ("i2t", DecAttnC)]
                + [(f"norm{i}_{p}", C.c_void_p) for i in (1, 2, 3, 4) for p in ("w", "b")] + [("lin1", LinC), ("lin2", LinC)])


class SamDecC(C.Structure):
    _fields_ = [("C", C.c_int), ("heads", C.c_int), ("depth", C.c_int), ("no_mask", C.c_void_p), ("key_pe", C.c_void_p),
                ("out_tokens", C.c_void_p), ("layers", DecLayerC * 4), ("final_attn", DecAttnC), ("norm_final_w", C.c_void_p),
                ("norm_final_b", C.c_void_p), ("up_ln_w", C.c_void_p), ("up_ln_b", C.c_void_p), ("up0", LinC), ("up1", LinC),
                ("hyper", LinC * 3), ("iou", LinC * 3)]


def _lin_c(w2, b):
    return LinC(w2.data_ptr(), b.data_ptr() if b is not None else None, w2.shape[0], w2.shape[1] // 2)


class SamDecodeStages:
    """``ivlm_sam_decode`` over the weights of a ``sam.SamMaskDecoder``."""

    def __init__(self, dec):
        self.d = dec
        attn = lambda a: DecAttnC(*[_lin_c(a[n].w, a[n].b) for n in ("q", "k", "v", "o")])
        w = SamDecC()
        w.C, w.heads, w.depth = dec.C, 8, len(dec.layers)
        w.no_mask, w.key_pe, w.out_tokens = dec.no_mask.data_ptr(), dec.key_pe.data_ptr(), dec.out_tokens.data_ptr()
        for i, L in enumerate(dec.layers):
            norms = [t for k in ("norm1", "norm2", "norm3", "norm4") for t in (L[k].w.data_ptr(), L[k].b.data_ptr())]
            w.layers[i] = DecLayerC(attn(L["self_attn"]), attn(L["t2i"]), attn(L["i2t"]), *norms, _lin_c(L["lin1"].w, L["lin1"].b),
                                    _lin_c(L["lin2"].w, L["lin2"].b))
        w.final_attn = attn(dec.final_attn)
        w.norm_final_w, w.norm_final_b = dec.norm_final.w.data_ptr(), dec.norm_final.b.data_ptr()
        w.up_ln_w, w.up_ln_b = dec.up_ln.w.data_ptr(), dec.up_ln.b.data_ptr()
        w.up0, w.up1 = _lin_c(dec.up0_w, dec.up0_b), _lin_c(dec.up1_w, dec.up1_b)
        for j in range(3):
            w.hyper[j] = _lin_c(dec.hyper0[j].w, dec.hyper0[j].b)
            w.iou[j] = _lin_c(dec.iou[j].w, dec.iou[j].b)
        self.w = w
        self.mlp_dim = dec.layers[0]["lin1"].w.shape[0]

    def __call__(self, image_embeddings, text_embeds):
        """image_embeddings fp32 [V, g*g, C], text_embeds fp32 [1, T, C] -> (low_res f32 [V,1,4g,4g], iou f32 [V,1])."""
        lib = _lib.load()
        V, HW, Cc = image_embeddings.shape
        g = self.d.grid
        emb = image_embeddings.to(torch.float32).contiguous()
        txt = text_embeds[0].to(torch.float32).contiguous()
        low = torch.empty(V, 4 * g, 4 * g, dtype=torch.float32, device=emb.device)
        iou = torch.empty(V, self.w.iou[2].n, dtype=torch.float32, device=emb.device)
        nbytes = lib.ivlm_sam_decode_workspace_bytes(V, g, Cc, txt.shape[0], self.mlp_dim)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)
        check(lib.ivlm_sam_decode(C.byref(self.w), V, g, txt.shape[0], emb.data_ptr(), txt.data_ptr(), low.data_ptr(),
                                  iou.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "sam_decode")
        return low.unsqueeze(1), iou[:, 0:1]
