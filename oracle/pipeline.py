"""End-to-end CPU oracle of InteractVLMForCausalLM.model_forward(inference=True) / evaluate().

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Follows model/InteractVLM.py:296-474 (teacher-forced)
and :510-638 (generate; here with a forced-token schedule + greedy argmax bookkeeping), composing
``oracle.nn`` and ``oracle.lift``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lift as L
from . import nn as O

SAM = "model.visual_model"
CLIPP = "model.vision_tower.vision_tower.vision_model"


def encode_images(w, cfg, images_clip):
    """llava_arch.py:93-96: CLIP penultimate patch features -> mm_projector."""
    c = cfg.clip
    f = O.clip_vision(w, CLIPP, images_clip, c.layers, c.heads, c.select_layer, c.patch, c.eps)
    return O.linear(w, "model.mm_projector", f)


def llm_hidden(w, cfg, input_ids, image_features):
    """One full-sequence forward -> last hidden state (post final norm) [T+255, H] for one sample."""
    emb = O.splice_image_features(w, input_ids, image_features)
    return O.llama(w, "model", emb[None], cfg.llama.layers, cfg.llama.heads, cfg.llama.eps, cfg.llama.theta)[0]


def greedy(w, cfg, images_clip, input_ids, max_new_tokens=32, eos_token_id=2):
    """The reference's free-running generation (model/InteractVLM.py:524-531: HF greedy search, num_beams = 1, with
    use_cache = False at :128, so that EVERY step is an uncached forward of the whole prefix with the image features spliced in -
    llava_arch.py:98-123 - followed by lm_head on the last position and argmax): full re-forward, argmax, append, stop on EOS or
    after max_new_tokens.  -> (output ids [L + n] (prompt + new tokens, like outputs.sequences), top-2 logit margins of the n steps
    [n] - a comparison with another implementation is only meaningful where the margin exceeds that implementation's tolerance -,
    the logits of every step [n, vocab])."""
    feat = encode_images(w, cfg, images_clip)[0]
    ids = input_ids.clone()
    margins, logits_all = [], []
    for _ in range(max_new_tokens):
        hidden = llm_hidden(w, cfg, ids, feat)
        logits = hidden[-1] @ w["lm_head.weight"].float().t()
        top2 = torch.topk(logits, 2).values
        margins.append(float(top2[0] - top2[1]))
        logits_all.append(logits)
        tok = int(torch.argmax(logits))
        ids = torch.cat([ids, torch.tensor([tok], dtype=ids.dtype)])
        if tok == eos_token_id:
            break
    return ids, torch.tensor(margins), torch.stack(logits_all)


def sam_embed(w, cfg, images):
    """get_visual_embs (InteractVLM.py:251-261): images [V,3,S,S] -> [V,256,g,g]."""
    s = cfg.sam
    return O.sam_image_encoder(w, SAM + ".image_encoder", images, s.depth, s.num_heads, s.global_attn_indexes,
                               s.window, s.patch)


def decode_masks(w, cfg, seg_emb, token, cam_params, image_embeddings, input_size, original_size):
    """pred_emb [n_seg,256] -> pred_masks [V,H,W] fp32 (InteractVLM.py:416-442 / 585-612)."""
    V = cfg.multiview_channels
    pe_cfg = dict(multiview_cam_cond=cfg.multiview_cam_cond, cam_encoder_type=cfg.cam_encoder_type,
                  multiview_channels=V, base_token_type=cfg.token_type.replace("-DifDe", ""),
                  hseg_token_idx=cfg.hseg_token_idx, oseg_token_idx=cfg.oseg_token_idx)
    emb = seg_emb.unsqueeze(1)
    if V > 1:
        emb = emb.repeat(1, V, 1)
    emb = O.process_embeddings(w, emb, cam_params, token, pe_cfg)
    g = cfg.sam.grid
    sparse, dense = O.prompt_encoder_text(w, SAM + ".prompt_encoder", emb, (g, g))
    pe = O.dense_pe(w, SAM + ".prompt_encoder", (g, g))
    low, iou = O.mask_decoder(w, SAM + ".mask_decoder", image_embeddings, pe, sparse, dense)
    masks = O.postprocess_masks(low, input_size, original_size, cfg.sam.img_size)
    return masks[:, 0], low, iou


def model_forward(w, cfg, images, images_clip, input_ids, cam_params, tables, input_size=None, original_size=None):
    """One sample (B=1), hcontact.  images [V,3,S,S], images_clip [1,3,224,224], input_ids [L] (one -200),
    tables=(vid,bary) -> dict(pred_masks [V,H,W], pred_contact [1,Nv], + taps)."""
    S = cfg.sam.img_size
    input_size = input_size or (S, S)
    original_size = original_size or (S, S)
    feat = encode_images(w, cfg, images_clip)[0]
    hidden = llm_hidden(w, cfg, input_ids, feat)
    seg_ids = [cfg.seg_token_idx]
    if cfg.token_type.replace("-DifDe", "") in ("Gen-Hu-Obj", "Gen-Int"):
        seg_ids += [cfg.hseg_token_idx, cfg.oseg_token_idx]
    rows = O.seg_rows(input_ids, seg_ids, cfg.img_emb_len, model_forward=True)
    seg_emb = O.text_hidden_fcs(w, hidden)[rows]
    k = int(rows.nonzero()[0]) - cfg.img_emb_len + 1
    token = int(input_ids[k]) if k > 0 else None
    emb = sam_embed(w, cfg, images)
    extra = {}
    dec_emb = emb
    if getattr(cfg, "use_fusion", False):  # ModifiedSAM.forward, InteractVLM.py:41-44 (llava_features = the whole sequence, :414,431)
        dec_emb = O.sam_fusion(w, "model.visual_model.fusion", emb, hidden[None])
        extra["fused_emb"] = dec_emb
    if getattr(cfg, "use_uncertainty", False):  # on the un-fused embeddings, InteractVLM.py:445-448
        unc = O.uncertainty_head(w, "model.visual_model.uncertainty", emb)
        extra["uncertainty_map"] = O.uncertainty_resize(unc, original_size)
    masks, low, iou = decode_masks(w, cfg, seg_emb, token, cam_params, dec_emb, input_size, original_size)
    vid, bary = tables
    contact, nviews = L.lift_mesh_soft(masks.numpy()[None], vid, bary, 6890)
    return dict(clip_feat=feat, hidden=hidden, seg_emb=seg_emb, sam_emb=emb, low_res=low, pred_masks=masks,
                pred_contact=torch.from_numpy(contact), nviews=nviews, **extra)


def model_forward_oafford(w, cfg, images, images_clip, input_ids, cam_params, point_maps, valid_mask,
                          input_size=None, original_size=None):
    """One 'oafford' sample with an 'HM' object view type (InteractVLM.py:452-456, components.py:289-347):
    pred masks get a sigmoid on the pixels whose gt mask is not IGNORE_LABEL (valid_mask [V,H,W] bool), then the
    per-view pixel->point maps (int [V,H,W], -1 = none) vote onto the 2048 points.
    -> dict(pred_masks [V,H,W], pred_afford [1, Np])."""
    S = cfg.sam.img_size
    input_size = input_size or (S, S)
    original_size = original_size or (S, S)
    feat = encode_images(w, cfg, images_clip)[0]
    hidden = llm_hidden(w, cfg, input_ids, feat)
    rows = O.seg_rows(input_ids, [cfg.seg_token_idx], cfg.img_emb_len, model_forward=True)
    seg_emb = O.text_hidden_fcs(w, hidden)[rows]
    k = int(rows.nonzero()[0]) - cfg.img_emb_len + 1
    token = int(input_ids[k]) if k > 0 else None
    emb = sam_embed(w, cfg, images)
    masks, low, iou = decode_masks(w, cfg, seg_emb, token, cam_params, emb, input_size, original_size)
    masks = torch.where(torch.as_tensor(valid_mask), torch.sigmoid(masks), masks)
    afford, _ = L.lift_points(masks.numpy()[None], point_maps[None], 2048)
    return dict(pred_masks=masks, pred_afford=torch.from_numpy(afford), low_res=low)
