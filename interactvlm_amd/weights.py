"""Parameter inventories (state-dict key -> shape) of the sub-models on the contact-inference path.

Key names are exactly the reference's state-dict names (``merge_lora_weights_and_save_hf_model.py``
saves everything except ``vision_tower.*``), so a released checkpoint maps 1:1 onto these specs and the
synthetic weights used for parity (``synth_weights``) are poured into the reference's own modules by key.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Tuple

Spec = Dict[str, Tuple[int, ...]]

SAM_PREFIX = "model.visual_model"
CLIP_PREFIX = "model.vision_tower.vision_tower.vision_model"


@dataclass
class SamEncCfg:  # build_sam.py:15-22,56-82 (ViT-H defaults)
    embed_dim: int = 1280
    depth: int = 32
    num_heads: int = 16
    global_attn_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    img_size: int = 1024
    patch: int = 16
    window: int = 14
    out_chans: int = 256
    mlp_ratio: int = 4

    @property
    def grid(self):
        return self.img_size // self.patch


@dataclass
class ClipCfg:  # openai/clip-vit-large-patch14
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    inter: int = 4096
    image_size: int = 224
    patch: int = 14
    select_layer: int = -2
    eps: float = 1e-5

    @property
    def tokens(self):
        return (self.image_size // self.patch) ** 2 + 1


@dataclass
class LlamaCfg:  # LLaMA-2-7B defaults (13B: hidden 5120, layers 40, heads 40, inter 13824)
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    inter: int = 11008
    vocab: int = 32003
    eps: float = 1e-5
    theta: float = 10000.0
    max_pos: int = 1024


@dataclass
class IvlmCfg:
    llama: LlamaCfg = field(default_factory=LlamaCfg)
    clip: ClipCfg = field(default_factory=ClipCfg)
    sam: SamEncCfg = field(default_factory=SamEncCfg)
    out_dim: int = 256
    img_emb_len: int = 255
    seg_token_idx: int = 32000
    hseg_token_idx: int = None
    oseg_token_idx: int = None
    im_start_idx: int = 32001
    im_end_idx: int = 32002
    token_type: str = "Gen"
    multiview_channels: int = 4
    multiview_cam_cond: bool = True
    cam_encoder_type: str = "vi_v1"
    hC_sam_view_type: str = "4MV-Z_Vitru"
    oC_sam_view_type: str = "4MV-Z_HM"
    hC_loss_weight: float = 1.0
    oC_loss_weight: float = 0.0
    # '-DifDe' checkpoints carry mask_decoder.*, human_mask_decoder.* and object_mask_decoder.*.  "reference" (default): the
    # reference's own inference construction - from_pretrained loads the three key sets into ALIASED modules (InteractVLM.py:30-32:
    # human_/object_mask_decoder ARE mask_decoder until initialize_separate_decoders, which evaluate.py:557-563 calls AFTER the load),
    # so all three end up holding the key set loaded last, object_mask_decoder.* (module order of ModifiedSAM) - what the reference
    # evaluates on the same checkpoint, hence the drop-in default.  "separate" (opt-in): each decoder holds ITS OWN tensors and
    # ModifiedSAM.forward's selection by dataset name (InteractVLM.py:46-52) picks among them - what the separately trained copies
    # are for.  The two differ only for a checkpoint whose copies differ; checkpoint.load_weights warns then.
    difde_load: str = "reference"
    use_fusion: bool = False       # LLaVASAMFusion head (InteractVLM.py:149; off in every released configuration)
    use_uncertainty: bool = False  # UncertaintyModule head (InteractVLM.py:150)


def _lin(spec: Spec, name: str, out: int, inp: int, bias: bool = True):
    spec[name + ".weight"] = (out, inp)
    if bias:
        spec[name + ".bias"] = (out,)


def _ln(spec: Spec, name: str, n: int):
    spec[name + ".weight"] = (n,)
    spec[name + ".bias"] = (n,)


def sam_encoder_spec(c: SamEncCfg, prefix: str = SAM_PREFIX + ".image_encoder") -> Spec:
    s: Spec = {}
    D = c.embed_dim
    hd = D // c.num_heads
    s[prefix + ".pos_embed"] = (1, c.grid, c.grid, D)
    s[prefix + ".patch_embed.proj.weight"] = (D, 3, c.patch, c.patch)
    s[prefix + ".patch_embed.proj.bias"] = (D,)
    for i in range(c.depth):
        p = f"{prefix}.blocks.{i}"
        side = c.grid if i in c.global_attn_indexes else c.window
        _ln(s, p + ".norm1", D)
        s[p + ".attn.rel_pos_h"] = (2 * side - 1, hd)
        s[p + ".attn.rel_pos_w"] = (2 * side - 1, hd)
        _lin(s, p + ".attn.qkv", 3 * D, D)
        _lin(s, p + ".attn.proj", D, D)
        _ln(s, p + ".norm2", D)
        _lin(s, p + ".mlp.lin1", c.mlp_ratio * D, D)
        _lin(s, p + ".mlp.lin2", D, c.mlp_ratio * D)
    s[prefix + ".neck.0.weight"] = (c.out_chans, D, 1, 1)
    _ln(s, prefix + ".neck.1", c.out_chans)
    s[prefix + ".neck.2.weight"] = (c.out_chans, c.out_chans, 3, 3)
    _ln(s, prefix + ".neck.3", c.out_chans)
    return s


def prompt_encoder_spec(prefix: str = SAM_PREFIX + ".prompt_encoder", embed_dim=256, mask_in_chans=16) -> Spec:
    s: Spec = {prefix + ".pe_layer.positional_encoding_gaussian_matrix": (2, embed_dim // 2)}
    for i in range(4):
        s[f"{prefix}.point_embeddings.{i}.weight"] = (1, embed_dim)
    s[prefix + ".not_a_point_embed.weight"] = (1, embed_dim)
    s[prefix + ".mask_downscaling.0.weight"] = (mask_in_chans // 4, 1, 2, 2)
    s[prefix + ".mask_downscaling.0.bias"] = (mask_in_chans // 4,)
    _ln(s, prefix + ".mask_downscaling.1", mask_in_chans // 4)
    s[prefix + ".mask_downscaling.3.weight"] = (mask_in_chans, mask_in_chans // 4, 2, 2)
    s[prefix + ".mask_downscaling.3.bias"] = (mask_in_chans,)
    _ln(s, prefix + ".mask_downscaling.4", mask_in_chans)
    s[prefix + ".mask_downscaling.6.weight"] = (embed_dim, mask_in_chans, 1, 1)
    s[prefix + ".mask_downscaling.6.bias"] = (embed_dim,)
    s[prefix + ".no_mask_embed.weight"] = (1, embed_dim)
    return s


def mask_decoder_spec(prefix: str = SAM_PREFIX + ".mask_decoder", C=256, mlp=2048, depth=2, n_mask=4) -> Spec:
    s: Spec = {}

    def attn(p, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            _lin(s, f"{p}.{n}", internal, C)
        _lin(s, p + ".out_proj", C, internal)

    for i in range(depth):
        p = f"{prefix}.transformer.layers.{i}"
        attn(p + ".self_attn", C)
        _ln(s, p + ".norm1", C)
        attn(p + ".cross_attn_token_to_image", C // 2)
        _ln(s, p + ".norm2", C)
        _lin(s, p + ".mlp.lin1", mlp, C)
        _lin(s, p + ".mlp.lin2", C, mlp)
        _ln(s, p + ".norm3", C)
        _ln(s, p + ".norm4", C)
        attn(p + ".cross_attn_image_to_token", C // 2)
    attn(prefix + ".transformer.final_attn_token_to_image", C // 2)
    _ln(s, prefix + ".transformer.norm_final_attn", C)
    s[prefix + ".iou_token.weight"] = (1, C)
    s[prefix + ".mask_tokens.weight"] = (n_mask, C)
    s[prefix + ".output_upscaling.0.weight"] = (C, C // 4, 2, 2)
    s[prefix + ".output_upscaling.0.bias"] = (C // 4,)
    _ln(s, prefix + ".output_upscaling.1", C // 4)
    s[prefix + ".output_upscaling.3.weight"] = (C // 4, C // 8, 2, 2)
    s[prefix + ".output_upscaling.3.bias"] = (C // 8,)
    for i in range(n_mask):
        p = f"{prefix}.output_hypernetworks_mlps.{i}"
        _lin(s, p + ".layers.0", C, C)
        _lin(s, p + ".layers.1", C, C)
        _lin(s, p + ".layers.2", C // 8, C)
    p = prefix + ".iou_prediction_head"
    _lin(s, p + ".layers.0", 256, C)
    _lin(s, p + ".layers.1", 256, 256)
    _lin(s, p + ".layers.2", n_mask, 256)
    return s


def clip_spec(c: ClipCfg, prefix: str = CLIP_PREFIX) -> Spec:
    s: Spec = {}
    e = prefix + ".embeddings"
    s[e + ".class_embedding"] = (c.hidden,)
    s[e + ".patch_embedding.weight"] = (c.hidden, 3, c.patch, c.patch)
    s[e + ".position_embedding.weight"] = (c.tokens, c.hidden)
    _ln(s, prefix + ".pre_layrnorm", c.hidden)
    for i in range(c.layers):
        p = f"{prefix}.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(s, f"{p}.self_attn.{n}", c.hidden, c.hidden)
        _ln(s, p + ".layer_norm1", c.hidden)
        _lin(s, p + ".mlp.fc1", c.inter, c.hidden)
        _lin(s, p + ".mlp.fc2", c.hidden, c.inter)
        _ln(s, p + ".layer_norm2", c.hidden)
    _ln(s, prefix + ".post_layernorm", c.hidden)
    return s


def llama_spec(c: LlamaCfg, prefix: str = "model") -> Spec:
    s: Spec = {prefix + ".embed_tokens.weight": (c.vocab, c.hidden)}
    for i in range(c.layers):
        p = f"{prefix}.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            _lin(s, f"{p}.self_attn.{n}", c.hidden, c.hidden, bias=False)
        _lin(s, p + ".mlp.gate_proj", c.inter, c.hidden, bias=False)
        _lin(s, p + ".mlp.up_proj", c.inter, c.hidden, bias=False)
        _lin(s, p + ".mlp.down_proj", c.hidden, c.inter, bias=False)
        s[p + ".input_layernorm.weight"] = (c.hidden,)
        s[p + ".post_attention_layernorm.weight"] = (c.hidden,)
    s[prefix + ".norm.weight"] = (c.hidden,)
    s["lm_head.weight"] = (c.vocab, c.hidden)
    return s


def cam_encoder_spec(kind: str, V: int = 4, prefix: str = "cam_pose_encoder") -> Spec:
    s: Spec = {}
    if kind == "simple":
        _lin(s, prefix + ".linear1", 256, 5)
    elif kind == "view_index":
        _lin(s, prefix + ".spatial_encoder.0", 256, 5)
        _lin(s, prefix + ".spatial_encoder.2", 256, 256)
        for v in range(V):
            _lin(s, f"{prefix}.view_transforms.{v}", 256, 256)
    elif kind == "vi_v1":
        _lin(s, prefix + ".spatial_encoder.0", 128, 5)
        _lin(s, prefix + ".spatial_encoder.2", 128, 128)
        for v in range(V):
            _lin(s, f"{prefix}.view_transforms.{v}", 256, 128)
    else:
        raise ValueError(kind)
    return s


def attention_splitter_spec(prefix: str = "attention_splitter") -> Spec:
    s: Spec = {}
    _lin(s, prefix + ".input_proj", 128, 256)
    for n in ("query_human", "query_object", "key", "value"):
        _lin(s, f"{prefix}.{n}", 128, 128)
    _lin(s, prefix + ".output_proj", 256, 128)
    return s


def uncertainty_spec(prefix: str = SAM_PREFIX + ".uncertainty") -> Spec:  # components.py:40-53
    s: Spec = {}
    _lin(s, prefix + ".linear1", 64, 256)
    _lin(s, prefix + ".linear2", 16, 64)
    _lin(s, prefix + ".linear3", 1, 16)
    return s


def fusion_spec(prefix: str = SAM_PREFIX + ".fusion", sam_dim=256, llava_dim=5120, fusion_dim=128) -> Spec:  # components.py:79-119
    s: Spec = {}
    _lin(s, prefix + ".sam_proj", fusion_dim, sam_dim)
    _lin(s, prefix + ".llava_proj", fusion_dim, llava_dim)
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        _lin(s, f"{prefix}.fusion.{n}", fusion_dim, fusion_dim)
    _lin(s, prefix + ".output_proj", sam_dim, fusion_dim)
    return s


def ivlm_spec(c: IvlmCfg) -> Spec:
    """Every tensor InteractVLMForCausalLM's state dict holds on the inference path."""
    s: Spec = {}
    s.update(llama_spec(c.llama))
    s.update(clip_spec(c.clip))
    _lin(s, "model.mm_projector", c.llama.hidden, c.clip.hidden)
    s.update(sam_encoder_spec(c.sam))
    s.update(prompt_encoder_spec())
    s.update(mask_decoder_spec())
    if "DifDe" in c.token_type:  # separately trained copies of the mask decoder (InteractVLM.py:114-121, selected at :46-52)
        s.update(mask_decoder_spec(SAM_PREFIX + ".human_mask_decoder"))
        s.update(mask_decoder_spec(SAM_PREFIX + ".object_mask_decoder"))
    _lin(s, "model.text_hidden_fcs.0.0", c.llama.hidden, c.llama.hidden)
    _lin(s, "model.text_hidden_fcs.0.2", c.out_dim, c.llama.hidden)
    if c.multiview_cam_cond:
        s.update(cam_encoder_spec(c.cam_encoder_type, c.multiview_channels))
    if c.token_type.replace("-DifDe", "") in ("Gen-Hu-Obj", "Gen-Int"):
        s.update(attention_splitter_spec())
    if c.use_fusion:  # (constructed with its defaults, InteractVLM.py:35: llava_embed_dim = 5120 whatever the language model is)
        s.update(fusion_spec())
    if c.use_uncertainty:
        s.update(uncertainty_spec())
    return s


def synth_weights(spec: Spec, seed: int = 0, dtype=None):
    """name -> torch tensor (fp32 unless dtype) of ``synth.synth_param``."""
    import torch

    from .synth import synth_param

    out = {}
    for k, shp in spec.items():
        t = torch.from_numpy(synth_param(k, shp, seed))
        out[k] = t if dtype is None else t.to(dtype)
    return out
