"""Checkpoint loader for released InteractVLM weights.

What the reference loads (run_demo.py:134-173, evaluate.py:540-573, utils/eval_utils.py:215-244):
  * ``<version>/``: an HF ``save_pretrained`` folder of ``InteractVLMForCausalLM`` written by
    merge_lora_weights_and_save_hf_model.py:152-161 - sharded ``pytorch_model-*.bin`` / ``model-*.safetensors`` (+ index
    json) or a single file, WITHOUT the ``vision_tower.*`` keys, plus ``config.json`` (LLaMA fields + the custom
    attributes) and ``pretrained_config.json`` (the training args; eval_utils.py overrides a fixed list of keys from
    ``config.json``);
  * the CLIP tower from ``openai/clip-vit-large-patch14`` (``model.config.vision_tower``), re-attached under
    ``model.vision_tower.vision_tower.`` by ``initialize_vision_modules``.
State-dict keys are used unchanged (``weights.py`` is the inventory); tensors stay on the CPU in their stored dtype and
are validated name by name and shape by shape against ``weights.ivlm_spec`` before anything is uploaded.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional

import torch

from . import weights as Wt

_HF_OVERRIDE_KEYS = ("hC_sam_view_type", "hC_question_type", "token_type", "cam_encoder_type", "multiview_cam_cond",
                     "multiview_channels", "img_emb_len")  # utils/eval_utils.py:224-228


class CheckpointError(RuntimeError):
    pass


def _read_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    return obj


def read_hf_state_dict(folder: str) -> Dict[str, torch.Tensor]:
    """All tensors of an HF ``save_pretrained`` folder (single file or sharded, safetensors or torch pickles)."""
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ip = os.path.join(folder, index)
        if os.path.exists(ip):
            with open(ip) as f:
                wm = json.load(f)["weight_map"]
            out: Dict[str, torch.Tensor] = {}
            for shard in sorted(set(wm.values())):
                part = _read_file(os.path.join(folder, shard))
                out.update({k: v for k, v in part.items() if wm.get(k) == shard})
            missing = set(wm) - set(out)
            if missing:
                raise CheckpointError(f"{folder}: index lists tensors absent from the shards: {sorted(missing)[:5]} ...")
            return out
    for single in ("model.safetensors", "pytorch_model.bin"):
        sp = os.path.join(folder, single)
        if os.path.exists(sp):
            return _read_file(sp)
    raise CheckpointError(f"{folder}: no model.safetensors / pytorch_model.bin (or their .index.json) found")


def read_clip_state_dict(folder: str, prefix: str = Wt.CLIP_PREFIX) -> Dict[str, torch.Tensor]:
    """CLIP vision tower of ``openai/clip-vit-large-patch14`` (a CLIPModel or CLIPVisionModel folder) under the key
    prefix the reference re-attaches it with (clip_encoder.py:22-29 inside ``model.vision_tower``)."""
    sd = read_hf_state_dict(folder)
    out = {}
    for k, v in sd.items():
        if k.startswith("vision_model."):
            out[prefix + k[len("vision_model"):]] = v
    if not out:
        raise CheckpointError(f"{folder}: no 'vision_model.*' tensors (not a CLIP checkpoint?)")
    return out


def infer_clip_cfg(state: Dict[str, torch.Tensor], prefix: str = Wt.CLIP_PREFIX) -> Wt.ClipCfg:
    """Tower dimensions from the tensors themselves (config.json only names the tower)."""
    pe = state[prefix + ".embeddings.patch_embedding.weight"]
    hidden, patch = int(pe.shape[0]), int(pe.shape[-1])
    n_tok = int(state[prefix + ".embeddings.position_embedding.weight"].shape[0])
    layers = 1 + max(int(k[len(prefix) + len(".encoder.layers."):].split(".")[0]) for k in state
                     if k.startswith(prefix + ".encoder.layers."))
    inter = int(state[prefix + ".encoder.layers.0.mlp.fc1.weight"].shape[0])
    grid = int(round((n_tok - 1) ** 0.5))
    return Wt.ClipCfg(hidden=hidden, layers=layers, heads=max(1, hidden // 64), inter=inter, image_size=grid * patch,
                      patch=patch)  # OpenAI CLIP ViTs: head dim 64


def infer_sam_cfg(state: Dict[str, torch.Tensor], prefix: str = Wt.SAM_PREFIX + ".image_encoder") -> Wt.SamEncCfg:
    pos = state[prefix + ".pos_embed"]  # [1, grid, grid, D]
    grid, dim = int(pos.shape[1]), int(pos.shape[-1])
    patch = int(state[prefix + ".patch_embed.proj.weight"].shape[-1])
    depth = 1 + max(int(k[len(prefix) + len(".blocks."):].split(".")[0]) for k in state if k.startswith(prefix + ".blocks."))
    rel = [state[f"{prefix}.blocks.{i}.attn.rel_pos_h"] for i in range(depth)]
    head_dim = int(rel[0].shape[1])
    glob = tuple(i for i, r in enumerate(rel) if int(r.shape[0]) == 2 * grid - 1)  # global blocks see the whole grid
    win = [(int(r.shape[0]) + 1) // 2 for i, r in enumerate(rel) if i not in glob]
    return Wt.SamEncCfg(embed_dim=dim, depth=depth, num_heads=dim // head_dim, global_attn_indexes=glob,
                        img_size=grid * patch, patch=patch, window=win[0] if win else 14,
                        out_chans=int(state[prefix + ".neck.0.weight"].shape[0]))


def config_from_hf(folder: str, tokenizer_ids: Optional[dict] = None,
                   state: Optional[Dict[str, torch.Tensor]] = None) -> Wt.IvlmCfg:
    """IvlmCfg from ``config.json`` (+ ``pretrained_config.json`` with the eval_utils override rule).
    tokenizer_ids: optional {'[SEG]': id, '[HSEG]': id, '[OSEG]': id, '<im_start>': id, '<im_end>': id} as produced by
    the tokenizer the caller loaded (run_demo.py:87-118); falls back to the ids stored in the config."""
    with open(os.path.join(folder, "config.json")) as f:
        hf = json.load(f)
    args = {}
    pc = os.path.join(folder, "pretrained_config.json")
    if os.path.exists(pc):
        with open(pc) as f:
            args = json.load(f)
        for k in _HF_OVERRIDE_KEYS:
            if k in hf:
                args[k] = hf[k]

    def get(name, default=None):
        return hf.get(name, args.get(name, default))

    heads = int(hf["num_attention_heads"])
    if int(hf.get("num_key_value_heads", heads)) != heads:
        raise CheckpointError("grouped-query attention checkpoints are not supported (LLaMA-2 7B/13B use MHA)")
    llama = Wt.LlamaCfg(hidden=int(hf["hidden_size"]), layers=int(hf["num_hidden_layers"]), heads=heads,
                        inter=int(hf["intermediate_size"]), vocab=int(hf["vocab_size"]),
                        eps=float(hf.get("rms_norm_eps", 1e-5)), theta=float(hf.get("rope_theta", 10000.0)),
                        max_pos=int(hf.get("max_position_embeddings", 4096)))
    tk = tokenizer_ids or {}
    towers = {}
    if state is not None:  # CLIP / SAM sizes are not in config.json: read them off the tensors
        towers = dict(clip=infer_clip_cfg(state), sam=infer_sam_cfg(state))
    cfg = Wt.IvlmCfg(
        llama=llama, **towers, out_dim=int(get("out_dim", 256)), img_emb_len=int(get("img_emb_len", 255)),
        seg_token_idx=int(tk.get("[SEG]", get("seg_token_idx", 32000))),
        hseg_token_idx=tk.get("[HSEG]", get("hseg_token_idx")), oseg_token_idx=tk.get("[OSEG]", get("oseg_token_idx")),
        im_start_idx=int(tk.get("<im_start>", get("im_start_idx", 32001))),
        im_end_idx=int(tk.get("<im_end>", get("im_end_idx", 32002))),
        token_type=str(get("token_type", "Gen")), multiview_channels=int(get("multiview_channels", 4)),
        multiview_cam_cond=bool(get("multiview_cam_cond", True)), cam_encoder_type=str(get("cam_encoder_type", "vi_v1")),
        hC_sam_view_type=str(get("hC_sam_view_type", "4MV-Z_Vitru")), oC_sam_view_type=str(get("oC_sam_view_type", "4MV-Z_HM")),
        hC_loss_weight=float(get("hC_loss_weight", 1.0)), oC_loss_weight=float(get("oC_loss_weight", 0.0)),
        use_fusion=bool(get("use_fusion", get("use_feat_fusion", False))), use_uncertainty=bool(get("use_uncertainty", False)))
    return cfg


# Tensors a checkpoint written by merge_lora_weights_and_save_hf_model.py:152-161 may carry besides the inference path's own
# (tests/golden/state_dict_keys.json is the reference's inventory): older transformers saved the rotary inv_freq buffers, and the
# optional fusion / uncertainty heads (ModifiedSAM.fusion / .uncertainty, InteractVLM.py:33-38) are off in every released configuration
# and part of the spec only when config.json switches them on (use_fusion / use_uncertainty, InteractVLM.py:149-150).  The '-DifDe' token types carry two more mask
# decoders: separately TRAINED deep copies (initialize_separate_decoders, InteractVLM.py:114-121, called from train.py:274 and
# evaluate.py:557/563) that ModifiedSAM.forward selects by dataset name (:46-52) - with a '-DifDe' configuration they are part of
# the spec (weights.ivlm_spec) and are loaded as decoders of their own; a non-DifDe configuration ignores them if present.
TOLERATED_PREFIXES = ("model.visual_model.uncertainty.", "model.visual_model.fusion.", "model.visual_model.human_mask_decoder.",
                      "model.visual_model.object_mask_decoder.")
TOLERATED_SUFFIXES = ("rotary_emb.inv_freq",)


def check_against_spec(state: Dict[str, torch.Tensor], cfg: Wt.IvlmCfg, ignore: Iterable[str] = TOLERATED_PREFIXES,
                       strict_extras: bool = False) -> None:
    """Every tensor of the inference path present with the right shape; raises CheckpointError listing what is wrong.
    Known extras (``ignore`` prefixes, TOLERATED_SUFFIXES) are skipped silently; any other unknown tensor is a warning (an error
    with strict_extras).  The '-DifDe' decoders are spec entries of a '-DifDe' configuration (they may differ from
    ``mask_decoder.*``: they are trained separately)."""
    import warnings

    spec = Wt.ivlm_spec(cfg)
    missing = [k for k in spec if k not in state]
    bad = [(k, tuple(state[k].shape), tuple(spec[k])) for k in spec if k in state and tuple(state[k].shape) != tuple(spec[k])]
    extra = [k for k in state if k not in spec and not any(k.startswith(p) for p in ignore)
             and not k.endswith(TOLERATED_SUFFIXES)]
    if extra and not strict_extras:
        warnings.warn(f"checkpoint carries {len(extra)} tensors outside the inference path (ignored): {extra[:6]}"
                      f"{' ...' if len(extra) > 6 else ''}")
        extra = []
    if missing or bad or extra:
        msg = [f"checkpoint does not match the {cfg.llama.layers}-layer / {cfg.cam_encoder_type} configuration:"]
        if missing:
            msg.append(f"  missing ({len(missing)}): {missing[:6]}{' ...' if len(missing) > 6 else ''}")
        if bad:
            msg.append(f"  shape mismatch ({len(bad)}): {bad[:4]}{' ...' if len(bad) > 4 else ''}")
        if extra:
            msg.append(f"  unexpected ({len(extra)}): {extra[:6]}{' ...' if len(extra) > 6 else ''}")
        raise CheckpointError("\n".join(msg))


def warn_if_difde_copies_differ(state, cfg) -> bool:
    """A '-DifDe' checkpoint whose three decoder key sets differ evaluates differently here (cfg.difde_load == "separate": every
    decoder its own tensors) and in the reference's inference construction (all three = object_mask_decoder.*, see
    weights.IvlmCfg.difde_load): say so once, and which semantics is applied."""
    import warnings

    if "DifDe" not in cfg.token_type:
        return False
    pre = Wt.SAM_PREFIX
    differ = []
    for k in state:
        if k.startswith(pre + ".mask_decoder."):
            tail = k[len(pre + ".mask_decoder."):]
            for other in ("human_mask_decoder", "object_mask_decoder"):
                o = state.get(f"{pre}.{other}.{tail}")
                if o is not None and not torch.equal(o, state[k]):
                    differ.append(other)
    if differ:
        warnings.warn(f"'-DifDe' checkpoint: {sorted(set(differ))} differ from mask_decoder.* - applying difde_load="
                      f"{cfg.difde_load!r} ({'each decoder its own tensors, selected by dataset name' if cfg.difde_load == 'separate' else 'all three decoders = object_mask_decoder.*, what the reference constructs at inference'}); "
                      f"set IvlmCfg.difde_load to choose")
    return bool(differ)


def load_weights(version_dir: str, clip_dir: str, tokenizer_ids: Optional[dict] = None,
                 ignore: Iterable[str] = TOLERATED_PREFIXES):
    """(cfg, state) ready for ``InteractVLMForCausalLM(cfg, state, device)``."""
    state = read_hf_state_dict(version_dir)
    state = {k: v for k, v in state.items() if "vision_tower" not in k}  # (never stored; guard against older dumps)
    state.update(read_clip_state_dict(clip_dir))
    try:
        cfg = config_from_hf(version_dir, tokenizer_ids, state)
    except KeyError as e:
        raise CheckpointError(f"{version_dir}: tensor {e} needed to size the towers is missing") from None
    check_against_spec(state, cfg, ignore)
    warn_if_difde_copies_differ(state, cfg)
    return cfg, state


def load_model(version_dir: str, clip_dir: str, device="cuda:0", tokenizer_ids: Optional[dict] = None, **model_kwargs):
    """The drop-in for ``InteractVLMForCausalLM.from_pretrained(version, vision_tower=...)`` (run_demo.py:134-136)."""
    from .model import InteractVLMForCausalLM

    cfg, state = load_weights(version_dir, clip_dir, tokenizer_ids)
    return InteractVLMForCausalLM(cfg, state, device, **model_kwargs)
