"""Writers of synthetic checkpoints in the layouts the reference's release tooling produces (test helper)."""
import json
import os

import torch

from interactvlm_amd import weights as Wt


def _write_version(folder, cfg, state, fmt, shards=3):
    os.makedirs(folder)
    c = cfg.llama
    hf = dict(hidden_size=c.hidden, num_hidden_layers=c.layers, num_attention_heads=c.heads, num_key_value_heads=c.heads,
              intermediate_size=c.inter, vocab_size=c.vocab, rms_norm_eps=c.eps, rope_theta=c.theta,
              max_position_embeddings=c.max_pos, img_emb_len=255, seg_token_idx=cfg.seg_token_idx, token_type="Gen",
              cam_encoder_type=cfg.cam_encoder_type, multiview_cam_cond=True, multiview_channels=4, out_dim=256,
              vision_tower="openai/clip-vit-large-patch14", hC_sam_view_type="4MV-Z_Vitru")
    json.dump(hf, open(os.path.join(folder, "config.json"), "w"))
    # training args: keys that config.json overrides (eval_utils.py:224-228) carry stale values here on purpose
    json.dump(dict(token_type="STALE", cam_encoder_type="STALE", oC_sam_view_type="4MV-Z_HM", hC_loss_weight=1.0,
                   oC_loss_weight=0.5, exp_name="x"), open(os.path.join(folder, "pretrained_config.json"), "w"))
    body = {k: v.to(torch.bfloat16) for k, v in state.items() if "vision_tower" not in k}
    keys = sorted(body)
    if fmt == "single_st":
        from safetensors.torch import save_file
        save_file(body, os.path.join(folder, "model.safetensors"))
        return
    ext, index = (("safetensors", "model.safetensors.index.json") if fmt == "sharded_st"
                  else ("bin", "pytorch_model.bin.index.json"))
    wm = {}
    for s in range(shards):
        part = {k: body[k] for k in keys[s::shards]}
        name = (f"model-{s + 1:05d}-of-{shards:05d}.safetensors" if ext == "safetensors"
                else f"pytorch_model-{s + 1:05d}-of-{shards:05d}.bin")
        if ext == "safetensors":
            from safetensors.torch import save_file
            save_file(part, os.path.join(folder, name))
        else:
            torch.save(part, os.path.join(folder, name))
        wm.update({k: name for k in part})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(folder, index), "w"))


def _write_clip(folder, state):
    from safetensors.torch import save_file

    os.makedirs(folder)
    pre = Wt.CLIP_PREFIX
    sd = {"vision_model" + k[len(pre):]: v.contiguous() for k, v in state.items() if k.startswith(pre)}
    sd["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)  # a full CLIPModel also has a text tower
    save_file(sd, os.path.join(folder, "model.safetensors"))
