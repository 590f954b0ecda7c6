"""Checkpoint loader: HF folders in the layouts merge_lora_weights_and_save_hf_model.py:152-161 writes
(sharded safetensors / sharded torch pickles / single file, no vision_tower keys) + a CLIP folder -> (cfg, state)."""
import json
import os

import numpy as np
import pytest
import torch

from interactvlm_amd import checkpoint as C
from interactvlm_amd import synthetic
from interactvlm_amd import weights as Wt


from _ckpt_util import _write_clip, _write_version  # noqa: E402


@pytest.mark.parametrize("fmt", ["sharded_st", "sharded_bin", "single_st"])
def test_load_released_layouts(tmp_path, fmt):
    cfg = synthetic.config_tiny()
    state = Wt.synth_weights(Wt.ivlm_spec(cfg))
    ver, clip = str(tmp_path / "interactvlm-3d-hcontact-damon"), str(tmp_path / "clip-vit-large-patch14")
    _write_version(ver, cfg, state, fmt)
    _write_clip(clip, state)
    cfg2, st2 = C.load_weights(ver, clip, tokenizer_ids={"[SEG]": 32000})
    assert cfg2.llama == cfg.llama and cfg2.cam_encoder_type == cfg.cam_encoder_type and cfg2.token_type == "Gen"
    assert cfg2.oC_loss_weight == 0.5 and cfg2.oC_sam_view_type == "4MV-Z_HM"  # only in pretrained_config.json
    assert set(st2) == set(Wt.ivlm_spec(cfg))
    for k in ("model.layers.1.mlp.down_proj.weight", "lm_head.weight", "model.visual_model.mask_decoder.iou_token.weight"):
        assert torch.equal(st2[k].float(), state[k].to(torch.bfloat16).float())
    k = Wt.CLIP_PREFIX + ".encoder.layers.0.self_attn.q_proj.weight"
    assert torch.equal(st2[k], state[k])


def test_mismatches_are_reported(tmp_path):
    cfg = synthetic.config_tiny()
    state = Wt.synth_weights(Wt.ivlm_spec(cfg))
    ver, clip = str(tmp_path / "v"), str(tmp_path / "clip")
    bad = dict(state)
    del bad["model.layers.0.self_attn.o_proj.weight"]
    bad["model.norm.weight"] = torch.zeros(7)
    bad["model.surprise.weight"] = torch.zeros(1)
    _write_version(ver, cfg, bad, "single_st")
    _write_clip(clip, state)
    with pytest.raises(C.CheckpointError) as e:
        C.load_weights(ver, clip)
    msg = str(e.value)
    assert "o_proj" in msg and "model.norm.weight" in msg and "surprise" in msg
    with pytest.raises(C.CheckpointError):
        C.read_hf_state_dict(str(tmp_path / "clip" / ".."))  # a folder without any model file
    with pytest.raises(C.CheckpointError):
        C.read_clip_state_dict(ver)  # not a CLIP checkpoint


def test_dispatch_choice_helpers_are_pure_host_logic():
    """Split-K choice and the rel-pos table concatenation run without a GPU (host logic of ops.py)."""
    import torch

    from interactvlm_amd import ops

    # LLaMA prefill o_proj / down_proj (few tiles, long K) split; big SAM GEMMs and SwiGLU / RMS-fused calls do not
    assert ops._splitk_choice(330, 4096, 4096, "none", None) > 1
    assert ops._splitk_choice(330, 4096, 11008, "none", None) > 1
    assert ops._splitk_choice(16384, 3840, 1280, "none", None) == 1
    assert ops._splitk_choice(330, 22016, 4096, "swiglu", None) == 1
    assert ops._splitk_choice(12, 4096, 4096, "none", None) == 1  # skinny MFMA kernel's territory
    th = torch.arange(27 * 8, dtype=torch.float32).reshape(27, 8).to(torch.bfloat16)
    tw = -th
    cat = ops.relpos_tables_cat(th, tw)
    assert cat.shape == (56, 8) and torch.equal(cat[:27], th) and torch.equal(cat[27:54], tw) and not cat[54:].any()
