"""InteractVLMForCausalLM — drop-in host facade of the contact-inference path on MI355X.

Same public surface as the reference's model/InteractVLM.py (forward / model_forward / evaluate /
get_visual_embs / process_embeddings, same argument names and result-dict keys), same
sub-module attribute names (``model.visual_model.{image_encoder,prompt_encoder,mask_decoder}``,
``human_3d_contact_predictor`` ...).  All arithmetic is in libivlm_hip.so; this file only sequences launches.

MI355X-first restructuring that leaves results unchanged:
  * CLIP runs once per image, the LLM decodes against a KV cache (reference: full re-forward incl. CLIP per
    generated token, InteractVLM.py:128,524-531);
  * text_hidden_fcs runs on the selected [SEG] rows only (reference: whole sequence, then boolean-mask select);
  * lift tables are resident in HBM as a vertex-major plan (reference: 150 MB H2D per view per call).
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import List, Optional

import torch

from . import ops
from .components import HumanContact3DPredictor, ObjectMeshContact3DPredictor, ObjectPCAfford3DPredictor
from .constants import IGNORE_LABEL, IMAGE_TOKEN_INDEX
from .heads import SamFusionHead, UncertaintyHead
from .llava import ClipTower, Llama
from .sam import SamImageEncoder, SamMaskDecoder, _Lin, _dev, postprocess_masks
from .weights import IvlmCfg

BF16 = torch.bfloat16
F32 = torch.float32


class _CamPoseEncoder:
    """CamPoseEncoder / ViewIndexCamPoseEncoder / VIv1CamPoseEncoder (components.py:491-572).
    The 5-wide first layer is zero-padded to K=8 (16-byte rows) for the streaming GEMV.  fp32 activations (V rows: the
    weight-streaming kernel multiplies bf16 weights with fp32 activations exactly)."""

    def __init__(self, w, kind, V, device, prefix="cam_pose_encoder"):
        self.kind, self.V = kind, V

        def lin(name, pad_k=None):
            W, b = w[name + ".weight"], w[name + ".bias"]
            if pad_k:
                W = torch.nn.functional.pad(W, (0, pad_k - W.shape[1]))
            return _dev(W, device), _dev(b, device)

        if kind == "simple":
            self.l1 = lin(prefix + ".linear1", 8)
        else:
            self.s0 = lin(prefix + ".spatial_encoder.0", 8)
            self.s2 = lin(prefix + ".spatial_encoder.2")
            self.views = [lin(f"{prefix}.view_transforms.{v}") for v in range(V)]

    def __call__(self, cam_params):
        """cam_params [V,5] -> view encodings fp32 [V,256] (row v = encoder(cam_params[v], view_idx=v))."""
        c = torch.nn.functional.pad(cam_params.to(F32), (0, 3)).contiguous()
        if self.kind == "simple":
            return ops.linear(c, self.l1[0], self.l1[1], act="relu", out_f32=True)
        h = ops.linear(c, self.s0[0], self.s0[1], act="relu", out_f32=True)
        # view_index: Linear-ReLU-Linear-Sigmoid, then per-view Linear; vi_v1: Linear-ReLU-Linear-ReLU, per-view Linear-Sigmoid
        vi = self.kind == "view_index"
        base = ops.linear(h, self.s2[0], self.s2[1], act="sigmoid" if vi else "relu", out_f32=True)
        enc = torch.empty(self.V, self.views[0][0].shape[0], dtype=F32, device=base.device)
        for v in range(self.V):
            ops.linear(base[v: v + 1], *self.views[v], act="none" if vi else "sigmoid", out=enc[v: v + 1])
        return enc


class InteractVLMForCausalLM:
    def __init__(self, config: IvlmCfg, weights: dict, device="cuda:0", lift_tables=None, metadata_root="./data",
                 max_len=1024, precision=None):
        c = self.config = config
        self.device = dev = torch.device(device)
        w = weights
        self.hC_sam_view_type, self.oC_sam_view_type = c.hC_sam_view_type, c.oC_sam_view_type
        self.hC_loss_weight, self.oC_loss_weight = c.hC_loss_weight, c.oC_loss_weight
        self.seg_token_idx, self.hseg_token_idx, self.oseg_token_idx = c.seg_token_idx, c.hseg_token_idx, c.oseg_token_idx
        self.token_type, self.img_emb_len = c.token_type, c.img_emb_len
        self.multiview_channels, self.multiview_cam_cond = c.multiview_channels, c.multiview_cam_cond
        self.cam_encoder_type = c.cam_encoder_type
        self.base_token_type = c.token_type.replace("-DifDe", "")
        # optional heads of ModifiedSAM (InteractVLM.py:33-38); off in every released config (scripts/run_train.sh:61-62)
        self.use_fusion, self.use_uncertainty = bool(c.use_fusion), bool(c.use_uncertainty)
        self.debug_taps = None  # set to a dict to record intermediate tensors (tests / diagnostics only)
        self.overlap_sam_encoder = True
        # HIP-graph replay of the decode step / CLIP tower (launch-bound on the host otherwise); IVLM_NO_GRAPHS=1 turns both
        # off (rocprofv3 --pmc passes crash on replayed graphs)
        self.graph_decode = not os.environ.get("IVLM_NO_GRAPHS")
        self.sam_after_prefill = bool(os.environ.get("IVLM_SAM_AFTER_PREFILL"))  # measured: 107.6 vs 106.6 ms - overlapping the decode instead of the prefill is not better
        self.packed_prefill = True  # generate_batch: prefill all prompts of a batch as one packed pass (rows independent)
        self.fused_lowres_lift = False  # measured slower than lifting the (cache-resident) full-res masks
        self._side_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._hi_stream = torch.cuda.Stream(device=dev, priority=-1) if dev.type == "cuda" else None
        self.prioritise_llm = False  # measured: no gain (the two streams time-share the CUs either way)

        self.vision_tower = ClipTower(w, c.clip, dev)
        if os.environ.get("IVLM_NO_GRAPHS"):
            self.vision_tower.use_graph = False
        self.mm_projector = _Lin(w, "model.mm_projector", dev)
        self.llm = Llama(w, c.llama, dev, max_len=max_len)
        self.text_hidden_fcs = (_Lin(w, "model.text_hidden_fcs.0.0", dev), _Lin(w, "model.text_hidden_fcs.0.2", dev))
        vm = SimpleNamespace()
        vm.image_encoder = SamImageEncoder(w, c.sam, dev)
        vm.mask_decoder = SamMaskDecoder(w, dev, grid=c.sam.grid)
        if os.environ.get("IVLM_NO_GRAPHS"):
            vm.mask_decoder.use_graph = False
            vm.image_encoder.use_graph = False
        vm.prompt_encoder = vm.mask_decoder  # text path of the prompt encoder is folded into the decoder object
        self.use_diff_decoder = "DifDe" in c.token_type
        if self.use_diff_decoder:  # separately trained decoder copies, picked per sample by dataset name (InteractVLM.py:46-52)
            assert c.difde_load in ("separate", "reference"), c.difde_load
            if c.difde_load == "reference":  # the reference's load-into-aliases-then-deepcopy: all three = the key set loaded last
                vm.mask_decoder = SamMaskDecoder(w, dev, grid=c.sam.grid, decoder="object_mask_decoder")
                vm.prompt_encoder = vm.mask_decoder
                if os.environ.get("IVLM_NO_GRAPHS"):
                    vm.mask_decoder.use_graph = False
            names = ("object_mask_decoder",) * 2 if c.difde_load == "reference" else ("human_mask_decoder", "object_mask_decoder")
            vm.human_mask_decoder = SamMaskDecoder(w, dev, grid=c.sam.grid, decoder=names[0])
            vm.object_mask_decoder = SamMaskDecoder(w, dev, grid=c.sam.grid, decoder=names[1])
            if os.environ.get("IVLM_NO_GRAPHS"):
                vm.human_mask_decoder.use_graph = vm.object_mask_decoder.use_graph = False
        if self.use_fusion:
            vm.fusion = SamFusionHead(w, dev)
        if self.use_uncertainty:
            vm.uncertainty = UncertaintyHead(w, dev, grid=c.sam.grid)
        vm.postprocess_masks = lambda m, input_size, original_size: postprocess_masks(
            m, input_size, original_size, c.sam.img_size)
        self.model = SimpleNamespace(visual_model=vm, text_hidden_fcs=self.text_hidden_fcs,
                                     mm_projector=self.mm_projector, vision_tower=self.vision_tower)
        self.lm_head = self.llm.lm_head
        self.human_3d_contact_predictor = self.object_3d_afford_predictor = self.object_3d_contact_predictor = None
        if c.hC_loss_weight > 0:
            self.human_3d_contact_predictor = HumanContact3DPredictor(
                c.hC_sam_view_type, c.multiview_channels, metadata_root=metadata_root, tables=lift_tables)
        if c.oC_loss_weight > 0:
            self.object_3d_afford_predictor = ObjectPCAfford3DPredictor(c.oC_sam_view_type, c.multiview_channels)
            self.object_3d_contact_predictor = ObjectMeshContact3DPredictor(c.oC_sam_view_type, c.multiview_channels)
        self.cam_pose_encoder = (_CamPoseEncoder(w, c.cam_encoder_type, c.multiview_channels, dev)
                                 if c.multiview_cam_cond else None)
        self.attention_splitter = None
        if self.base_token_type in ("Gen-Hu-Obj", "Gen-Int"):
            self.attention_splitter = {n: _Lin(w, "attention_splitter." + n, dev) for n in
                                       ("input_proj", "query_human", "query_object", "key", "value", "output_proj")}
        # start-up precision mode (its weight copies are built here: a checkpoint with a weight outside fp16's range raises NOW -
        # construct it with precision="bf16" or a parity mode instead)
        self.set_precision(precision or os.environ.get("IVLM_PRECISION", "default"))

    # ------------------------------------------------------------------------------------------
    def eval(self):
        return self

    # Precision modes (what the MFMA operands of the three towers are; the residual streams, the decode step, text_hidden_fcs and the
    # mask decoder have fp32 activations in every mode, the weights are the checkpoint's bf16 values in every mode).
    # "default": IEEE fp16 operands in ONE MFMA pass (fp16 copies of the bf16 weights: exact) - an fp16 operand carries an eighth of
    #   the bf16 rounding error at the same matrix-core rate - with SAM's q path exact (q = W_q . norm1 on hi + lo halves, its lo half
    #   in the rel-pos terms, which amplify q's rounding 6 x more than Q.K^T does; fp32 rel-pos terms).  Holds the north star's 1e-3
    #   on per-vertex probabilities at the real depth with a 2 x margin (4 - 6e-4 against the fp32 oracle over seeds and shapes:
    #   bench.py parity_vs_oracle_full_depth, tools/diag_f16.py) at ~5 % over the bf16 path.
    # "bf16": bf16 operands (the precision class of the reference's own bf16 GPU model; rounds 8 x coarser: 6e-3 .. 1e-2 at the real
    #   depth).  The fastest mode; NOT within 1e-3.  Also what the fp8 variant builds on.
    # "parity": no activation is ever rounded - CLIP, the LLaMA prefill and the SAM ViT-H encoder carry every MFMA operand as
    #   hi + lo bf16 halves (fp32-activation GEMMs / attention on the bf16 matrix cores at 2-3 x the MFMA work), K / V are cached as
    #   hi + lo planes: 8e-6 against the oracle (threshold sets exactly equal).
    # "parity-fast": "parity" with the encoder's MLP on fp16 operands (the only operands below fp32-equivalent precision): 2.4e-4.
    # ("f16": the default without the exact q path - 8 - 9e-4, i.e. inside 1e-3 without margin; kept for diagnostics, not listed.)
    precision_modes = ("default", "bf16", "parity-fast", "parity")
    precision = "default"

    def set_precision(self, mode):
        assert mode in self.precision_modes + ("f16",), mode
        self.precision = mode
        lang = {"parity": "parity", "parity-fast": "parity", "bf16": "default"}.get(mode, "f16")  # (tower-level names)
        self.vision_tower.precision = lang
        self.llm.set_precision(lang)
        enc = self.model.visual_model.image_encoder
        enc.precision = "default" if mode == "bf16" else "parity"  # ("parity" = the site-driven forward of the encoder)
        enc.parity_sites = {"parity": enc.PARITY_SITES, "f16": enc.SITES_F16, "default": enc.SITES_F16Q}.get(mode, enc.PARITY_SITES_FAST)
        # the weight copies this mode reads are built NOW (a failure - memory, a weight outside fp16's range - surfaces here, not in
        # the first forward), the ones it does not read are released (default mode: the bf16 originals of the LLaMA matrices, which
        # the fp16 prefill copies + the lossless 12-bit decode planes replace: 36 -> 23 GB of language-model weights for 7B)
        if self.device.type == "cuda" and not self.fp8:
            self.llm.prepare()
            if self.free_unused_weights:
                self.llm.release_unused()

    free_unused_weights = os.environ.get("IVLM_KEEP_ALL_WEIGHTS", "0") != "1"

    def resident_weight_bytes(self):
        """Language-model weight bytes resident in HBM, by form (`hbm_resident_gb` of the bench line)."""
        return self.llm.resident_bytes()

    # fp8 variant (BASELINE.json configs[4], opt-in; never a parity claim): e4m3 operands for the GEMMs of the SAM ViT-H encoder,
    # the CLIP tower and the LLaMA prefill, e4m3 WEIGHTS for the batch-1 decode linears.  Activation scales are calibrated on the
    # inputs given HERE and then fixed: evaluate other images afterwards.
    def enable_fp8(self, images_clip, images, input_ids):
        """images_clip [1,3,h,w], images [1,V,3,S,S], input_ids [1,L] (prompt, ideally with a typical answer appended)."""
        self._precision_before_fp8 = self.precision
        self.set_precision("bf16")  # (the e4m3 paths replace the bf16-operand launches)
        dev = self.device
        self.model.visual_model.image_encoder.enable_fp8(images[0].to(dev))
        self.vision_tower.enable_fp8(images_clip.to(dev))
        feats = self.encode_images(images_clip)[0]
        self.llm.enable_fp8(self._input_embeds(input_ids[0].to(dev), feats))
        self.fp8 = True

    def disable_fp8(self):
        self.model.visual_model.image_encoder.fp8 = False
        self.vision_tower.fp8 = False
        self.llm.disable_fp8()
        self.fp8 = False
        self.set_precision(getattr(self, "_precision_before_fp8", "default"))

    fp8 = False
    # evaluate / evaluate_batch: recompute a non-finite result of an fp16-operand mode with bf16 operands (IVLM_NONFINITE_GUARD=0: off)
    nonfinite_guard = os.environ.get("IVLM_NONFINITE_GUARD", "1") != "0"

    def _guard_applies(self):
        return (self.nonfinite_guard and not self.fp8 and self.precision in ("default", "f16", "parity-fast")
                and not getattr(self, "_in_guard", False))

    def _guarded(self, fn, args, kwargs, supplied_embeddings=None):
        """fp16 operands have 5 exponent bits (max 65504; bf16: 8, as fp32), and the kernels do not clamp: an activation outside
        that range becomes inf and everything downstream NaN.  One flag per call - the tower outputs (LLaMA hidden states, SAM
        embeddings) and the contacts are all finite - read back where the caller would read the result anyway: a call that fails it
        is recomputed in the `bf16` mode (same arithmetic, fp32's exponent range; the decode step on the bf16 weights, whose fp32
        activations have the full fp32 range - the packed planes stage x * 2^64) and reported.  Embeddings the CALLER supplied
        (`image_embeddings=`) cannot be fixed by recomputing: non-finite ones raise."""
        self._in_guard = True
        self._finite_flags = []
        try:
            out = fn(*args, **kwargs)
            if not self._guard_ok(out, supplied_embeddings):
                out = self._recompute_bf16(lambda: fn(*args, **kwargs))
            return out
        finally:
            self._in_guard = False

    def _guard_ok(self, out, supplied_embeddings=None):
        """the one flag of a guarded call (tower outputs collected by _decode_sample + every contact tensor of the result finite);
        non-finite embeddings that the CALLER supplied raise"""
        outs = out if isinstance(out, list) else [out]
        flags = self._finite_flags + [torch.isfinite(o[k]).all() for o in outs
                                      for k in ("pred_contact_3d", "pred_human_3d_contact", "pred_object_3d_contact",
                                                "pred_object_3d_afford") if o.get(k) is not None]
        if not flags or bool(torch.stack(flags).all()):
            return True
        if supplied_embeddings is not None:
            embs = supplied_embeddings if isinstance(supplied_embeddings, (list, tuple)) else [supplied_embeddings]
            if not all(bool(torch.isfinite(e).all()) for e in embs):
                raise ops.IvlmError("image_embeddings passed by the caller are not finite (an fp16-mode encoder pass that "
                                    "overflowed?): re-encode them, e.g. model.precompute_visual_embs(views) - which checks")
        return False

    def _recompute_bf16(self, fn):
        import warnings

        warnings.warn(f"non-finite contacts in precision mode {self.precision!r} (an activation left fp16's exponent "
                      "range): this call is recomputed with bf16 operands; consider model.set_precision('bf16')")
        # The mode switch is TEMPORARY: nothing is released for it (the fp16 copies / packed planes of `mode` stay, the bf16
        # matrices are rebuilt from the planes and kept until the next explicit set_precision) - a fallback must not churn tens of GB
        # of weights, least of all under a pipelined evaluate_batch whose next chunk is already enqueued on the copies of `mode`
        # (ADVICE r5).  In that deferred mode a fallback also SERIALISES the pipeline: the recomputation is enqueued behind the
        # next chunk's work and that chunk's captured decode graphs are re-captured after the switch back.
        mode, packed, free = self.precision, self.llm.decode_packed, self.free_unused_weights
        self.llm.decode_packed = False
        self.free_unused_weights = False
        try:
            self.set_precision("bf16")
            out = fn()
        finally:
            self.llm.decode_packed = packed
            try:
                self.set_precision(mode)
            finally:
                self.free_unused_weights = free
        for o in (out if isinstance(out, list) else [out]):
            o["recomputed_in_bf16"] = True
        return out

    def _finite_or_bf16(self, fn):
        """tensor-returning encoder entry points under the fp16 exponent-range guard: a non-finite result is recomputed with bf16
        operands (one flag read back per call)"""
        out = fn()
        if self._guard_applies() and not bool(torch.isfinite(out).all()):
            import warnings

            warnings.warn(f"non-finite SAM embeddings in precision mode {self.precision!r}: recomputed with bf16 operands")
            mode = self.precision
            self._in_guard = True
            self.set_precision("bf16")
            try:
                out = fn()
            finally:
                self.set_precision(mode)
                self._in_guard = False
        return out

    def get_visual_embs(self, pixel_values):
        """[B,V,3,S,S] -> image embeddings; returned in the reference's [B,V,256,g,g] shape (a strided view of the
        channels-last buffer the decoder consumes) — InteractVLM.py:251-261."""
        B, V = pixel_values.shape[:2]
        g = self.config.sam.grid
        emb = self._finite_or_bf16(lambda: self.model.visual_model.image_encoder(
            pixel_values.reshape((B * V,) + tuple(pixel_values.shape[2:]))))
        return emb.view(B, V, g, g, -1).permute(0, 1, 4, 2, 3)

    def precompute_visual_embs(self, images_views):
        """[V,3,S,S] -> channels-last SAM embeddings [V, g*g, 256] to pass as evaluate(image_embeddings=...)."""
        return self._finite_or_bf16(lambda: self.model.visual_model.image_encoder(images_views.to(self.device)))

    def forward(self, **kwargs):
        if "past_key_values" in kwargs:  # InteractVLM.py:263-266: the HF causal-LM forward (what HF generate() calls per step)
            return self._causal_lm_forward(**kwargs)
        return self.model_forward(**kwargs)

    @torch.no_grad()
    def _causal_lm_forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                           use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None):
        """LlavaLlamaForCausalLM.forward (llava_llama.py:55-135) for ONE sequence, inference only: logits, the last hidden state
        (eval-mode `hidden_states`, :121-124) and a cache handle.  past_key_values None (or use_cache False, the reference's own
        setting: InteractVLM.py:128): the whole sequence from position 0, `images` [1,3,h,w] spliced in at the image placeholder
        (llava_arch.py:98-123).  past_key_values = the handle a previous call returned: only the LAST id is new and is decoded
        against this model's KV cache.  The cache lives in the model (one sequence at a time), the handle is its length."""
        if labels is not None or inputs_embeds is not None or output_attentions:
            raise NotImplementedError("inference-only route: no labels / inputs_embeds / attention maps")
        ids = input_ids.reshape(-1).to(self.device)
        if past_key_values is None:
            if images is not None and bool((ids == IMAGE_TOKEN_INDEX).any()):
                x = self._input_embeds(ids, self.encode_images(images)[0])
            else:
                x = self.llm.embed_ids(ids.to(torch.int32).contiguous())
            pos0 = 0
        else:
            pos0 = int(past_key_values.length)
            x = self.llm.embed_ids(ids[-1:].to(torch.int32).contiguous())
        if pos0 + x.shape[0] > self.llm.max_len:
            raise ops.IvlmError(f"sequence of {pos0 + x.shape[0]} positions exceeds the KV cache (max_len={self.llm.max_len})")
        h = self.llm.forward(x, pos0)
        logits = self.llm.logits(h)  # fp32 rows (> 16 rows: ONE tile GEMM over hi + lo operands - lm_head is streamed once)
        cache = SimpleNamespace(length=pos0 + x.shape[0]) if (use_cache or past_key_values is not None) else None
        return SimpleNamespace(loss=None, logits=logits[None], past_key_values=cache, hidden_states=h[None], attentions=None)

    def process_embeddings(self, embedding, cam_params, token):
        """[n_seg,V,256] view conditioning (InteractVLM.py:268-294)."""
        n_seg, V, C = embedding.shape
        if self.multiview_cam_cond:
            enc = self.cam_pose_encoder(cam_params.to(self.device))  # [V,256]
            if self.cam_encoder_type == "simple":
                embedding = ops.add_rows(embedding.reshape(n_seg * V, C).contiguous(), enc).view(n_seg, V, C)
            else:
                embedding = ops.add_rows(embedding.reshape(n_seg * V, C).contiguous(), enc, op="mul").view(n_seg, V, C)
        if self.base_token_type == "Gen":
            return embedding
        if token == self.hseg_token_idx or token == self.oseg_token_idx:
            return self._split(embedding, human=(token == self.hseg_token_idx))
        return embedding

    def _split(self, x, human):
        """AttentionSplitter (components.py:155-193) on fp32 [n,V,256]: one softmax over V keys of width 128."""
        a = self.attention_splitter
        n, V, C = x.shape
        lin = lambda L, t: ops.linear(t, L.w, L.b, out_f32=True)  # n*V <= 16 fp32 rows: exact products
        xp = lin(a["input_proj"], x.reshape(n * V, C).contiguous())
        k, v = lin(a["key"], xp), lin(a["value"], xp)
        q = lin(a["query_human" if human else "query_object"], xp)
        hd = q.shape[-1]
        one_head = lambda t: t.view(n, 1, V, hd)  # a single head of width 128
        o = ops.attention_f32(one_head(q), one_head(k), one_head(v), hd ** -0.5)
        return lin(a["output_proj"], o.reshape(n * V, hd)).view(n, V, C)

    # ------------------------------------------------------------------------------------------
    def _input_embeds(self, ids_row, image_features):
        """prepare_inputs_labels_for_multimodal, mm_use_im_start_end branch (llava_arch.py:185-208):
        embed_tokens gather with the single IMAGE_TOKEN_INDEX replaced by the 256 projected CLIP rows."""
        pos = int((ids_row.cpu() == IMAGE_TOKEN_INDEX).nonzero()[0])  # (host ids: no device round trip behind the CLIP launch)
        ids = ids_row.to(self.device)
        n_img = image_features.shape[0]
        L = ids.numel()
        x = torch.empty(L - 1 + n_img, self.config.llama.hidden, dtype=F32, device=self.device)  # fp32 residual stream
        idx = ids.clamp(min=0).to(torch.int32)
        self.llm.embed_ids(idx[:pos].contiguous(), out=x[:pos])
        ops.gather_rows(image_features, out=x[pos: pos + n_img])
        if L - pos - 1 > 0:
            self.llm.embed_ids(idx[pos + 1:].contiguous(), out=x[pos + n_img:])
        return x

    def encode_images(self, images_clip):
        """llava_arch.py:93-96."""
        f = self.vision_tower(images_clip.to(self.device))
        B, T, C = f.shape  # ("parity" precision: C = 2 * hidden, [hi | lo] rows)
        return self.mm_projector(f.reshape(B * T, C), out_f32=True,
                                 a_split=self.vision_tower.precision in ("parity", "f16")).view(B, T, -1)  # fp32: rows of the LLM's input stream

    def _seg_token_ids(self):
        ids = [self.seg_token_idx]
        if self.base_token_type in ("Gen-Hu-Obj", "Gen-Int"):
            ids += [self.hseg_token_idx, self.oseg_token_idx]
        return [i for i in ids if i is not None]

    def _seg_rows(self, ids, extra_false_col):
        """Boolean row mask over the (len(ids) - 1 [+1]) + img_emb_len hidden rows (InteractVLM.py:331-341/545-549).
        Computed on the HOST (the ids of evaluate() are host values; a device tensor is read back once): no device round trip
        between the decode loop and the mask decoder - the tail of evaluate() is enqueued while the last decode steps still run."""
        ids = ids.cpu()
        m = torch.zeros_like(ids, dtype=torch.bool)
        for s in self._seg_token_ids():
            m |= ids == s
        m = m[1:]
        if extra_false_col:
            m = torch.cat([m, torch.zeros(1, dtype=torch.bool)])
        return torch.cat([torch.zeros(self.img_emb_len, dtype=torch.bool), m])

    def _mask_decoder_for(self, ds_name):
        """ModifiedSAM.forward's decoder choice (InteractVLM.py:46-54): with '-DifDe' the human decoder serves 'hcontact' samples,
        the object decoder 'oafford' / 'ocontact', the shared one anything else."""
        vm = self.model.visual_model
        if self.use_diff_decoder and ds_name is not None:
            if "hcontact" in ds_name:
                return vm.human_mask_decoder
            if "oafford" in ds_name or "ocontact" in ds_name:
                return vm.object_mask_decoder
        return vm.mask_decoder

    def _decode_sample(self, hidden, rows_mask, ids, cam_params, image_embeddings, input_size, original_size, ds_name=None,
                       sigmoid_gt=None, llava_features=None):
        """[SEG] rows -> pred_mask [V,H,W] fp32 for one sample (InteractVLM.py:416-442 / 585-612).  llava_features: the hidden
        rows the fusion head attends to when ``use_fusion`` (ModifiedSAM.forward, InteractVLM.py:41-44)."""
        if getattr(self, "_in_guard", False):  # (evaluate / evaluate_batch under the fp16 exponent-range guard: see _guarded)
            self._finite_flags.append(torch.isfinite(hidden).all() & torch.isfinite(image_embeddings).all())
        rows = rows_mask.cpu().nonzero().flatten()  # (host mask: see _seg_rows)
        ids = ids.cpu()
        V = self.multiview_channels
        if rows.numel() == 0:
            return torch.zeros((0,) + tuple(original_size), dtype=torch.float32, device=self.device), None
        if self.use_fusion:
            image_embeddings = self.model.visual_model.fusion(image_embeddings, llava_features)
        sel = hidden[rows.to(hidden.device)].contiguous()
        if sel.shape[0] > 16:
            raise ops.IvlmError("more than 16 [SEG] rows in one sample")
        # [n_seg, 256] fp32: a few fp32 rows through the weight-streaming kernels (exact bf16-weight x fp32 products)
        emb = self.text_hidden_fcs[1](self.text_hidden_fcs[0](sel, act="relu", out_f32=True), out_f32=True)
        if self.debug_taps is not None:
            self.debug_taps.update(hidden=hidden, seg_emb=emb, sam_emb=image_embeddings)
        k = int(rows[0]) - self.img_emb_len + 1
        token = int(ids[k]) if k > 0 else None
        emb = emb.unsqueeze(1)
        if V > 1:
            emb = emb.repeat(1, V, 1)
        emb = self.process_embeddings(emb, cam_params, token)
        low, iou = self._mask_decoder_for(ds_name)(image_embeddings, emb)
        if self.debug_taps is not None:
            self.debug_taps.update(prompt_emb=emb, low_res=low, iou=iou)
        self._last_low = (low, tuple(input_size), tuple(original_size))
        if sigmoid_gt is not None:  # 'oafford' + 'HM' views: sigmoid where the gt mask is labelled, in the postprocess kernel
            gt = sigmoid_gt.to(self.device, F32).reshape(low.shape[0], 1, *tuple(original_size)).contiguous()
            return postprocess_masks(low, input_size, original_size, self.config.sam.img_size, sigmoid_gt=gt,
                                     ignore_label=float(IGNORE_LABEL))[:, 0], iou
        return postprocess_masks(low, input_size, original_size, self.config.sam.img_size)[:, 0], iou

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def model_forward(self, images, images_clip, input_ids, labels=None, attention_masks=None, offset=None,
                      masks_list=None, label_list=None, gt_contact_3d_list=None, cam_params=None, resize_list=None,
                      ds_name_list=None, mask_paths_list=None, inference=False, **kwargs):
        """Teacher-forced single pass (InteractVLM.py:296-474); inference=True only (training is out of scope)."""
        if not inference:
            raise NotImplementedError("training (loss) path is out of scope of the inference hot path")
        if self._guard_applies():  # (fp16 exponent range: see _guarded)
            return self._guarded(self.model_forward, (images, images_clip, input_ids, labels, attention_masks, offset, masks_list,
                                                      label_list, gt_contact_3d_list, cam_params, resize_list, ds_name_list,
                                                      mask_paths_list, inference), kwargs)
        B = images.shape[0]
        assert offset is None or B == len(offset) - 1
        assert images_clip.shape[0] == 1 or images_clip.shape[0] == B
        feats = self.encode_images(images_clip)
        if self.debug_taps is not None:
            self.debug_taps["clip_feat"] = feats
        emb_sam = self.model.visual_model.image_encoder(
            images.to(self.device).reshape((B * images.shape[1],) + tuple(images.shape[2:])))
        V = images.shape[1]
        emb_sam = emb_sam.view(B, V, emb_sam.shape[1], emb_sam.shape[2])
        pred_masks, gt_masks, uncertainty_maps = [], [], []
        for i in range(B):
            ids = input_ids[i].to(self.device)
            x = self._input_embeds(ids, feats[i if feats.shape[0] > 1 else 0])
            hidden = self.llm.forward(x, 0)
            rows = self._seg_rows(ids, extra_false_col=True)
            osz = tuple(label_list[i].shape[-2:]) if label_list is not None else tuple(resize_list[i])
            ds_name = ds_name_list[i] if ds_name_list else "hcontact"
            gt = masks_list[i][:, 0] if masks_list is not None else None
            # InteractVLM.py:452-456: 'oafford' samples with 'HM' object views get a sigmoid on the labelled pixels
            hm = gt is not None and "oafford" in ds_name and "HM" in (self.oC_sam_view_type or "")
            # use_fusion: the sample's whole last-layer sequence (llava_features[i].unsqueeze(0), InteractVLM.py:414,431)
            pm, _ = self._decode_sample(hidden, rows, ids, cam_params[i], emb_sam[i], resize_list[i], osz, ds_name=ds_name,
                                        sigmoid_gt=gt if hm else None, llava_features=hidden if self.use_fusion else None)
            pred_masks.append(pm)
            gt_masks.append(gt)
            if self.use_uncertainty:  # on the un-fused embeddings, resized to the label size (InteractVLM.py:445-448)
                uncertainty_maps.append(self.model.visual_model.uncertainty.resized(emb_sam[i], osz).squeeze(0))
        ds_name_list = ds_name_list or ["hcontact"] * B
        result = {"gt_masks": gt_masks, "pred_masks": pred_masks}
        if self.use_uncertainty:
            result["uncertainty_maps"] = uncertainty_maps
        if self.hC_loss_weight > 0:
            result["pred_human_3d_contact"] = self.human_3d_contact_predictor(pred_masks, ds_name_list)
        if self.oC_loss_weight > 0:
            result["pred_object_3d_contact"] = self.object_3d_contact_predictor(pred_masks, ds_name_list, mask_paths_list)
            result["pred_object_3d_afford"] = self.object_3d_afford_predictor(pred_masks, ds_name_list, mask_paths_list)
        return result

    def _id_ring(self, n):
        """pinned host int32 array of >= n generated ids (written by asynchronous device-to-host copies, read one step late)"""
        r = getattr(self, "_ring", None)
        if r is None or r.numel() < n:
            r = self._ring = torch.empty(max(n, 64), dtype=torch.int32).pin_memory()
        return r

    @torch.no_grad()
    def generate(self, images_clip, input_ids, max_new_tokens=32, eos_token_id=2, forced_new_tokens=None,
                 after_prefill=None):
        """Greedy search with a KV cache for ONE sequence.  Returns (output_ids [1, L+n], hidden [L+n-1+255, H]).
        forced_new_tokens (extension for weight-free benchmarking): feed these ids instead of the argmax (the
        argmax/lm_head work is still done every step), like the reference's inference_type='forward'."""
        feats = self.encode_images(images_clip)[0]
        ids = input_ids[0]
        x = self._input_embeds(ids, feats)
        T0 = x.shape[0]
        n_max = len(forced_new_tokens) if forced_new_tokens is not None else max_new_tokens
        # KV-cache capacity (constructor argument max_len, default 1024 = 330 prompt positions + run_demo's 512 new tokens
        # with room to spare): never generate past it
        n_max = min(n_max, self.llm.max_len - T0)
        if n_max <= 0:
            raise ops.IvlmError(f"prompt of {T0} positions does not fit the KV cache (max_len={self.llm.max_len})")
        if forced_new_tokens is not None:
            forced_new_tokens = list(forced_new_tokens)[:n_max]
        hidden_all = torch.empty(T0 + n_max, self.config.llama.hidden, dtype=F32, device=self.device)
        h = self.llm.forward(x, 0)
        hidden_all[:T0].copy_(h)
        if after_prefill is not None:
            after_prefill()  # evaluate(): the SAM encoder is enqueued here, between the prefill and the decode loop
        new_ids = []
        last = h[T0 - 1: T0]
        pos = T0
        forced_dev = None
        if forced_new_tokens is not None:
            forced_dev = torch.tensor([int(t) for t in forced_new_tokens], dtype=torch.int32, device=self.device)
        self.last_argmax = []
        use_graph = self.graph_decode and not ops.TIMER.enabled and not torch.cuda.is_current_stream_capturing()
        if use_graph:
            # one decode step = one replay of a captured HIP graph (embed -> 32 layers -> norm -> lm_head -> argmax, position
            # read from device memory); the host issues 3 launches per token instead of ~165
            dg = self.llm.decode_graph()
            dg["pos"].fill_(T0)
            dg["pos64"].fill_(T0)
            fz = dg.get("fused")
            if fz is not None:
                fz["step"].zero_()
                fz["counters"].zero_()
                fz["status"].zero_()
            nxt = ops.argmax(self.llm.logits(last))
            if forced_new_tokens is not None:
                for step in range(n_max):
                    self.last_argmax.append(nxt)
                    new_ids.append(int(forced_new_tokens[step]))
                    if new_ids[-1] == eos_token_id or step == n_max - 1:
                        break
                    dg["tok"].copy_(forced_dev[step: step + 1])
                    dg["graph"].replay()
                    hidden_all[pos: pos + 1].copy_(dg["hidden"])
                    nxt = dg["nxt"].clone()
                    pos += 1
            else:
                # Free-running greedy search WITHOUT a host round trip per token (reference loop: InteractVLM.py:524-531,
                # stop on EOS or max_new_tokens).  The argmax of step s stays on the device and is the token of step s + 1
                # (`tok.copy_(nxt)`: device to device); every id is also copied - asynchronously - into a pinned host array.  The host
                # keeps ONE replay queued ahead of the one it is waiting for: before it enqueues step s + 1 it waits for the event of
                # step s - 1 and reads id s - 1 from the pinned array - so the GPU never idles between replays, and when id k turns
                # out to be EOS exactly one speculative step (the one that consumed EOS) has been enqueued: its hidden row and KV
                # row lie beyond the returned length and are dropped.
                ring = self._id_ring(n_max)
                evs = [torch.cuda.Event() for _ in range(n_max)]
                cur = torch.cuda.current_stream(self.device)
                ids_dev = torch.empty(n_max, dtype=torch.int32, device=self.device)
                ids_dev[0:1].copy_(nxt)
                ring[0:1].copy_(nxt, non_blocking=True)
                evs[0].record(cur)
                n_tok, enq = None, 0  # n_tok: number of new ids once known; enq: decode steps enqueued
                for step in range(n_max):
                    # (a) id `step - 1` (one step late): stop at EOS
                    if step >= 1:
                        evs[step - 1].synchronize()
                        if int(ring[step - 1]) == eos_token_id:
                            n_tok = step
                            break
                    if step == n_max - 1:
                        break
                    # (b) enqueue the decode step that consumes id `step` and produces id `step + 1`
                    dg["tok"].copy_(ids_dev[step: step + 1])
                    dg["graph"].replay()
                    hidden_all[pos + enq: pos + enq + 1].copy_(dg["hidden"])
                    ids_dev[step + 1: step + 2].copy_(dg["nxt"])
                    ring[step + 1: step + 2].copy_(dg["nxt"], non_blocking=True)
                    evs[step + 1].record(cur)
                    enq += 1
                if n_tok is None:  # no EOS among ids 0 .. n_max - 2: the last id decides nothing (max_new_tokens reached)
                    evs[n_max - 1].synchronize()
                    n_tok = n_max
                new_ids = [int(t) for t in ring[:n_tok].tolist()]
                self.last_argmax = [ids_dev[i: i + 1] for i in range(n_tok)]
                pos += n_tok - 1  # decode steps whose hidden rows count (a speculative step past EOS is dropped)
            if fz is not None and int(fz["status"].item()) != 0:
                # a bounded device-side wait of the fused attention + o_proj launch expired (its blocks were not co-resident,
                # e.g. a third stream holding the CUs): drop to the two-launch path for good and redo this generation
                self.llm.fuse_attn_oproj = False
                self.llm._dgraph = None
                return self.generate(images_clip, input_ids, max_new_tokens, eos_token_id, forced_new_tokens, None)
            out_ids = torch.cat([ids.cpu(), torch.tensor(new_ids, dtype=ids.dtype)])[None]
            return out_ids, hidden_all[:pos]
        for step in range(n_max):
            nxt = ops.argmax(self.llm.logits(last))  # int32 [1] on device
            self.last_argmax.append(nxt)
            if forced_new_tokens is not None:
                tok = int(forced_new_tokens[step])
                tok_t = forced_dev[step: step + 1]
            else:
                tok = int(nxt.item())
                tok_t = nxt
            new_ids.append(tok)
            if tok == eos_token_id or step == n_max - 1:
                break
            e = self.llm.embed_ids(tok_t)
            last = self.llm.forward(e, pos)
            hidden_all[pos: pos + 1].copy_(last)
            pos += 1
        out_ids = torch.cat([ids.cpu(), torch.tensor(new_ids, dtype=ids.dtype)])[None]
        return out_ids, hidden_all[:pos]

    @torch.no_grad()
    # ---- B images per call (BASELINE.json configs[2]: 8 images per GPU) -----------------------------------------------
    def generate_batch(self, images_clip, input_ids_list, max_new_tokens=32, eos_token_id=2, forced_new_tokens=None):
        """Greedy search for B sequences at once (extension: the reference's evaluate() is batch 1, evaluate.py:479).
        The prompts are prefilled one by one into their own KV-cache slab, then every decode step streams the weights ONCE
        for all B new tokens.  Per sequence the arithmetic is that of ``generate`` (same kernels, row-independent), so the
        outputs equal B separate calls.  forced_new_tokens: one list per sequence (or one shared list).
        -> [(output_ids [1, L+n], hidden [L+n-1+255, H])] * B."""
        B = len(input_ids_list)
        if B > 16:  # (checked before any work: the weight-streaming decode kernels take at most 16 rows per step)
            raise ops.IvlmError(f"generate_batch: {B} sequences per call, at most 16 (evaluate_batch chunks larger batches itself)")
        dev = self.device
        feats = self.encode_images(images_clip)
        kc, vc = self.llm.batch_cache(B)
        lo = self.llm.batch_cache_lo(B) if self.llm.precision == "parity" else None
        # one image for all prompts (configs[4]: a human-contact and an object prompt about the same picture): ONE CLIP pass
        shared = feats.shape[0] == 1 and B > 1
        xs = [self._input_embeds(input_ids_list[b].reshape(-1), feats[0 if shared else b]) for b in range(B)]
        T0 = [int(x.shape[0]) for x in xs]
        if forced_new_tokens is not None:
            if not isinstance(forced_new_tokens[0], (list, tuple)):
                forced_new_tokens = [list(forced_new_tokens)] * B
            n_seq = [len(f) for f in forced_new_tokens]
        else:
            n_seq = [max_new_tokens] * B
        n_seq = [min(n, self.llm.max_len - t) for n, t in zip(n_seq, T0)]
        if min(n_seq) <= 0:
            raise ops.IvlmError(f"a prompt of {max(T0)} positions does not fit the KV cache (max_len={self.llm.max_len})")
        n_max = max(n_seq)
        hidden_all = torch.empty(B, max(T0) + n_max, self.config.llama.hidden, dtype=F32, device=dev)
        last = torch.empty(B, self.config.llama.hidden, dtype=F32, device=dev)
        if self.packed_prefill and B > 1:  # the B prompts in one pass over the weights
            hs = self.llm.forward_packed(xs, kc, vc, lo)
        else:
            hs = [self.llm.forward(xs[b], 0, cache=(kc[:, b], vc[:, b]) + ((lo[0][:, b], lo[1][:, b]) if lo else ()))
                  for b in range(B)]
        for b, h in enumerate(hs):
            hidden_all[b, : T0[b]].copy_(h)
            last[b].copy_(h[T0[b] - 1])
        forced_dev = None
        if forced_new_tokens is not None:
            pad = [list(map(int, f[:n])) + [eos_token_id] * (n_max - n) for f, n in zip(forced_new_tokens, n_seq)]
            forced_dev = torch.tensor(pad, dtype=torch.int32, device=dev).t().contiguous()  # [n_max, B]
        use_graph = self.graph_decode and not ops.TIMER.enabled and not torch.cuda.is_current_stream_capturing()
        pos_t = torch.tensor(T0, dtype=torch.int32, device=dev)
        if use_graph:
            dg = self.llm.decode_graph_batch(B)
            dg["pos"].copy_(pos_t)
        rows = torch.arange(B, device=dev)
        nxt = ops.argmax(self.llm.logits(last))
        new_ids = [[] for _ in range(B)]
        done = [False] * B

        def absorb(toks):  # the ids of one step, in order: append to the sequences still running, mark those that stop
            for b in range(B):
                if not done[b]:
                    new_ids[b].append(int(toks[b]))
                    if toks[b] == eos_token_id or len(new_ids[b]) >= n_seq[b]:
                        done[b] = True

        if use_graph and forced_dev is None:
            # free-running: no host round trip per token (see generate): ids stay on the device, a pinned host copy is read one
            # step late, one speculative step at most is enqueued after the last sequence has stopped
            ring = self._id_ring(n_max * B)[: n_max * B].view(n_max, B)
            evs = [torch.cuda.Event() for _ in range(n_max)]
            cur = torch.cuda.current_stream(dev)
            ids_dev = torch.empty(n_max, B, dtype=torch.int32, device=dev)
            ids_dev[0].copy_(nxt)
            ring[0].copy_(nxt, non_blocking=True)
            evs[0].record(cur)
            absorbed = 0
            for step in range(n_max):
                if step >= 1:
                    evs[step - 1].synchronize()
                    absorb(ring[step - 1].tolist())
                    absorbed = step
                    if all(done):
                        break
                if step == n_max - 1:
                    break
                dg["tok"].copy_(ids_dev[step])
                idx = dg["pos"].to(torch.int64)  # positions BEFORE the step's += 1
                dg["graph"].replay()
                hidden_all[rows, idx] = dg["hidden"]
                ids_dev[step + 1].copy_(dg["nxt"])
                ring[step + 1].copy_(dg["nxt"], non_blocking=True)
                evs[step + 1].record(cur)
            if not all(done):
                evs[n_max - 1].synchronize()
                for st_ in range(absorbed, n_max):
                    absorb(ring[st_].tolist())
            n_steps = 0  # (the stepping loop below is skipped)
        else:
            n_steps = n_max
        for step in range(n_steps):
            if forced_dev is not None:
                tok_t = forced_dev[step]
                toks = [pad[b][step] for b in range(B)]
            else:
                tok_t = nxt
                toks = nxt.tolist()  # (eager launches: the host is the bottleneck anyway)
            absorb(toks)
            if all(done):
                break
            # (finished sequences keep stepping - their rows are ignored; a sequence whose position has reached the end of its
            #  cache slab is skipped by the attention kernel: nothing is appended past Tmax)
            if use_graph:
                dg["tok"].copy_(tok_t)
                idx = dg["pos"].to(torch.int64)  # positions BEFORE the step's += 1
                dg["graph"].replay()
                h, nxt = dg["hidden"], dg["nxt"].clone()
            else:
                h = self.llm.decode_step_batch(self.llm.embed_ids(tok_t.contiguous()), pos_t, kc, vc, lo)
                idx = pos_t.to(torch.int64)
                nxt = ops.argmax(self.llm.logits(h))
                pos_t = pos_t + 1
            hidden_all[rows, idx] = h
        out = []
        for b in range(B):
            ids = input_ids_list[b].reshape(-1).cpu()
            out_ids = torch.cat([ids, torch.tensor(new_ids[b], dtype=ids.dtype)])[None]
            out.append((out_ids, hidden_all[b, : T0[b] + len(new_ids[b]) - 1]))
        return out

    def evaluate_batch(self, images_clip, images, input_ids_list, cam_params, resize_list, original_size_list,
                       contact_type="hcontact", max_new_tokens=32, forced_new_tokens=None, eos_token_id=2,
                       lift2d_dict_path=None, image_embeddings=None, deferred=False):
        """``evaluate`` for B images in one call: images_clip [B,3,h,w], images [B,V,3,S,S], one prompt per image.
        The SAM encoder of every image runs on the side stream while the B sequences decode together; the masks of all
        images are lifted in one launch.  -> [{'output_ids','pred_masks','pred_contact_3d'}] * B, each equal to what
        ``evaluate`` returns for that image alone.
        images_clip [1,3,h,w] with B > 1 prompts = B questions about ONE picture (one CLIP pass; configs[4]: a human-contact
        prompt over the body renders and an object prompt over the object renders); contact_type and lift2d_dict_path
        may then be lists, one entry per prompt.
        image_embeddings: pre-computed SAM embeddings, one [V, g*g, 256] tensor for all samples (the four
        canonical body renders of hcontact are the same for every image) or a list of B; ``images`` is then not encoded."""
        # deferred=True: everything that does not depend on the encoder is ENQUEUED now (SAM encoder on the side stream, CLIP +
        # prefill + the batched decode loop on the caller's stream) and a function is returned that enqueues the tail (mask decoders,
        # lift) and returns the results.  A caller with several chunks begins chunk c + 1 before it finishes chunk c
        # (dist.evaluate_sharded, and the > 16 split below): the side stream then runs encoder after encoder without waiting for
        # the ~35 ms of tiny mask-decoder launches of the previous chunk, which run under it.
        args = (images_clip, images, input_ids_list, cam_params, resize_list, original_size_list, contact_type, max_new_tokens,
                forced_new_tokens, eos_token_id, lift2d_dict_path, image_embeddings)
        B = len(input_ids_list)
        if B > 16:  # larger batches run as consecutive (pipelined) calls of <= 16 sequences (the decode kernels' row limit)
            if images_clip.shape[0] == 1:
                raise ops.IvlmError("evaluate_batch: more than 16 prompts about ONE picture are not supported")
            subs = []
            for lo in range(0, B, 16):
                hi = min(lo + 16, B)
                fn = forced_new_tokens
                if fn is not None and isinstance(fn[0], (list, tuple)):
                    fn = fn[lo:hi]
                emb = image_embeddings[lo:hi] if isinstance(image_embeddings, (list, tuple)) else image_embeddings
                subs.append((images_clip[lo:hi], None if images is None else images[lo:hi], input_ids_list[lo:hi], cam_params[lo:hi],
                             resize_list[lo:hi], original_size_list[lo:hi],
                             contact_type if isinstance(contact_type, str) else contact_type[lo:hi], max_new_tokens, fn, eos_token_id,
                             lift2d_dict_path[lo:hi] if isinstance(lift2d_dict_path, (list, tuple)) else lift2d_dict_path, emb))

            def finish_all():
                outs, pending = [], None
                for sub in subs:
                    nxt_ = self.evaluate_batch(*sub, deferred=True)
                    if pending is not None:
                        outs.extend(pending())
                    pending = nxt_
                outs.extend(pending())
                return outs
            return finish_all if deferred else finish_all()
        guard = self._guard_applies()  # (fp16 exponent range: see _guarded)
        st = self._evaluate_batch_begin(*args)

        def finish():
            if not guard:
                return self._evaluate_batch_finish(st)
            self._in_guard = True
            self._finite_flags = []
            try:
                outs = self._evaluate_batch_finish(st)
                if not self._guard_ok(outs, image_embeddings):
                    outs = self._recompute_bf16(lambda: self.evaluate_batch(*args))
                return outs
            finally:
                self._in_guard = False
        return finish if deferred else finish()

    def _evaluate_batch_begin(self, images_clip, images, input_ids_list, cam_params, resize_list, original_size_list,
                              contact_type, max_new_tokens, forced_new_tokens, eos_token_id, lift2d_dict_path, image_embeddings):
        B = len(input_ids_list)
        main = torch.cuda.current_stream(self.device)
        ev = allv = None
        if image_embeddings is not None:
            embs = list(image_embeddings) if isinstance(image_embeddings, (list, tuple)) else [image_embeddings] * B
            gens = self.generate_batch(images_clip, input_ids_list, max_new_tokens, eos_token_id, forced_new_tokens)
        else:
            side = self._side_stream if self.overlap_sam_encoder else main
            embs = []
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                # ALL B x V views in one encoder pass: the GEMMs see M = B * V * 4096 rows (proj / mlp2 of 8 images: 2560
                # tiles of 256^2 = 10 full rounds of the 256 CUs instead of 8 x (1 + a quarter-filled one)), an eighth of the
                # launches.  Row-wise the arithmetic is that of the batch-1 pass.
                V_ = images.shape[1]
                allv = self.model.visual_model.image_encoder(
                    images.to(self.device).reshape((B * V_,) + tuple(images.shape[2:])))
                embs = [allv[b * V_: (b + 1) * V_] for b in range(B)]
                if side is not main:
                    ev = torch.cuda.Event()
                    ev.record(side)
            gens = self.generate_batch(images_clip, input_ids_list, max_new_tokens, eos_token_id, forced_new_tokens)
        return dict(B=B, gens=gens, embs=embs, ev=ev, allv=allv, cam_params=cam_params, resize_list=resize_list,
                    original_size_list=original_size_list, contact_type=contact_type, lift2d_dict_path=lift2d_dict_path)

    def _evaluate_batch_finish(self, st):
        B, gens, embs = st["B"], st["gens"], st["embs"]
        cam_params, resize_list, original_size_list = st["cam_params"], st["resize_list"], st["original_size_list"]
        contact_type, lift2d_dict_path = st["contact_type"], st["lift2d_dict_path"]
        main = torch.cuda.current_stream(self.device)
        if st["ev"] is not None:
            main.wait_event(st["ev"])
            st["allv"].record_stream(main)
        outs, lows = [], []
        ctypes = [contact_type] * B if isinstance(contact_type, str) else list(contact_type)
        for b, (output_ids, hidden) in enumerate(gens):
            rows = self._seg_rows(output_ids[0], extra_false_col=False)
            pm, _ = self._decode_sample(hidden, rows, output_ids[0], cam_params[b], embs[b], resize_list[b],
                                        original_size_list[b], ds_name=ctypes[b], llava_features=self._eval_llava_features(hidden, 0))
            outs.append({"output_ids": output_ids, "pred_masks": [pm], "pred_contact_3d": None})
            if self.use_uncertainty:
                outs[-1]["uncertainty_maps"] = [self._eval_uncertainty(embs[b], original_size_list[b])]
            lows.append(pm)
        paths = lift2d_dict_path if isinstance(lift2d_dict_path, (list, tuple)) else [lift2d_dict_path] * B
        have = [b for b in range(B) if lows[b].shape[0] > 0]
        hum = [b for b in have if self.hC_loss_weight > 0 and "hcontact" in ctypes[b]]
        if hum:  # all body lifts in one launch (same tables)
            pc = self.human_3d_contact_predictor([lows[b] for b in hum])
            for i, b in enumerate(hum):
                outs[b]["pred_contact_3d"] = pc[i: i + 1]
        for b in have:
            if b in hum:
                continue
            if (self.oC_loss_weight > 0 and "ocontact" in ctypes[b]) or "oafford" in ctypes[b]:
                # every object has its own mesh and tables (components.py:463: batch size 1): one lift per image
                outs[b]["pred_contact_3d"] = self.object_3d_contact_predictor(
                    [lows[b]], ds_names=["ocontact"], lift2d_dict_path=paths[b])
        return outs

    def evaluate(self, images_clip, images, input_ids, cam_params, resize_list, original_size_list,
                 lift2d_dict_path=None, contact_type="hcontact", max_new_tokens=32, tokenizer=None,
                 forced_new_tokens=None, eos_token_id=2, image_embeddings=None):
        """Generate -> [SEG] hidden state -> SAM decode -> lift (InteractVLM.py:510-638).

        image_embeddings (extension): pre-computed SAM embeddings [V, g*g, 256] of ``images``
        (``precompute_visual_embs``).  For hcontact the SAM inputs are the SAME four canonical body renders for
        every sample (run_demo.py:279-292, datasets/hcontact_3d.py:268-271), so they can be encoded once."""
        assert input_ids.shape[0] == 1, "the reference only ever calls evaluate with batch 1 (evaluate.py:479)"
        if self._guard_applies():
            return self._guarded(self.evaluate, (images_clip, images, input_ids, cam_params, resize_list, original_size_list,
                                                 lift2d_dict_path, contact_type, max_new_tokens, tokenizer, forced_new_tokens,
                                                 eos_token_id, image_embeddings), {}, supplied_embeddings=image_embeddings)
        # The SAM ViT-H encoder (MFMA-bound, ~60 ms) does not depend on the language model (CLIP -> prefill -> decode:
        # HBM-bound weight streaming that leaves the matrix cores idle): run it on a second HIP stream and join
        # before the mask decoder.  The reference runs them back to back (InteractVLM.py:524-531, 578).
        side = None
        box = {"emb": image_embeddings, "ev": None}
        main = torch.cuda.current_stream(self.device)

        def launch_sam():
            side.wait_stream(main)  # inputs were produced on the caller's stream
            with torch.cuda.stream(side):
                box["emb"] = self.model.visual_model.image_encoder(images[0].to(self.device))
                box["ev"] = torch.cuda.Event()
                box["ev"].record(side)

        after_prefill = None
        if image_embeddings is None and self.overlap_sam_encoder:
            side = self._side_stream
            if self.sam_after_prefill:
                after_prefill = launch_sam  # MFMA-bound encoder overlaps the HBM-bound decode, not the MFMA-bound prefill
            else:
                launch_sam()
        # ... and the language path (the critical path) on a HIGH-priority stream, so that whenever CU slots free up
        # its workgroups are dispatched ahead of the encoder's
        hi = self._hi_stream if (side is not None and self.prioritise_llm) else None
        if hi is not None:
            hi.wait_stream(main)
            with torch.cuda.stream(hi):
                out = self._evaluate_tail(images_clip, images, input_ids, cam_params, resize_list, original_size_list,
                                          lift2d_dict_path, contact_type, max_new_tokens, forced_new_tokens,
                                          eos_token_id, box, side, after_prefill)
            main.wait_stream(hi)
            for tns in list(out["pred_masks"]) + [out["pred_contact_3d"]]:
                if tns is not None:
                    tns.record_stream(main)
            return out
        return self._evaluate_tail(images_clip, images, input_ids, cam_params, resize_list, original_size_list,
                                   lift2d_dict_path, contact_type, max_new_tokens, forced_new_tokens, eos_token_id,
                                   box, side, after_prefill)

    def _evaluate_tail(self, images_clip, images, input_ids, cam_params, resize_list, original_size_list,
                       lift2d_dict_path, contact_type, max_new_tokens, forced_new_tokens, eos_token_id,
                       box, side, after_prefill):
        output_ids, hidden = self.generate(images_clip, input_ids, max_new_tokens, eos_token_id, forced_new_tokens,
                                           after_prefill=after_prefill)
        rows = self._seg_rows(output_ids[0], extra_false_col=False)
        image_embeddings = box["emb"]
        if side is not None:
            torch.cuda.current_stream(self.device).wait_event(box["ev"])
            image_embeddings.record_stream(torch.cuda.current_stream(self.device))
        if image_embeddings is None:
            image_embeddings = self.model.visual_model.image_encoder(images[0].to(self.device))
        pm, _ = self._decode_sample(hidden, rows, output_ids[0], cam_params[0], image_embeddings, resize_list[0],
                                    original_size_list[0], ds_name=contact_type, llava_features=self._eval_llava_features(hidden, 0))
        pred_masks = [pm]
        pred_contact_3d = None
        if pred_masks[0].shape[0] > 0:
            if self.hC_loss_weight > 0 and "hcontact" in contact_type:
                low, isz, osz = self._last_low
                if self.fused_lowres_lift and low.shape[0] == self.multiview_channels:
                    # same numbers as lifting pred_masks, without re-reading the full-resolution masks (8f-1)
                    pred_contact_3d = self.human_3d_contact_predictor.forward_lowres(
                        [low], isz, osz, self.config.sam.img_size)
                else:
                    pred_contact_3d = self.human_3d_contact_predictor(pred_masks)
            elif (self.oC_loss_weight > 0 and "ocontact" in contact_type) or "oafford" in contact_type:
                # same operator precedence as the reference (InteractVLM.py:626): 'oafford' always takes the mesh lift
                pred_contact_3d = self.object_3d_contact_predictor(pred_masks, ds_names=["ocontact"],
                                                                   lift2d_dict_path=lift2d_dict_path)
        result = {"output_ids": output_ids, "pred_masks": pred_masks, "pred_contact_3d": pred_contact_3d}
        if self.use_uncertainty:
            result["uncertainty_maps"] = [self._eval_uncertainty(image_embeddings, original_size_list[0])]
        return result

    def _eval_llava_features(self, hidden, i):
        """evaluate()'s argument to the fusion head: ``output_hidden_states[-1]`` is the LAST sample's [T, hidden] sequence and
        ``llava_features[i].unsqueeze(0)`` its ROW i as [1, hidden] (InteractVLM.py:583,601) - one key / value position, which the
        head can only deal to a single view (components.py:94-95 raises for multiview_channels > 1; so does SamFusionHead)."""
        return hidden[i: i + 1] if self.use_fusion else None

    def _eval_uncertainty(self, image_embeddings, original_size):
        """evaluate()'s uncertainty map.  The reference hands UncertaintyModule a 5-D tensor there (image_embeddings[i].unsqueeze(0)
        with multi-view embeddings, InteractVLM.py:615 -> the permute of components.py:62 raises); this returns what its
        model_forward branch computes for the same sample (InteractVLM.py:445-448) instead of failing."""
        return self.model.visual_model.uncertainty.resized(image_embeddings, original_size).squeeze(0)
