"""Pins the PyTorch-CPU oracle of the neural part (oracle/nn.py, oracle/pipeline.py) to golden vectors
produced by running the reference's own modules (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from interactvlm_amd import synth
from interactvlm_amd import weights as Wt
from interactvlm_amd.weights import SAM_PREFIX
from oracle import nn as O
from oracle import pipeline as P

torch.set_grad_enabled(False)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("V", [4, 1])
def test_sam_decoder_chain(golden_dir, V):
    d = _g(golden_dir, f"sam_decoder_V{V}.npz")
    w = Wt.synth_weights({**Wt.prompt_encoder_spec(), **Wt.mask_decoder_spec()})
    emb = torch.from_numpy(synth.synth_normal(f"samdec/image_emb/{V}", (V, 256, 64, 64), 1.0, 0))
    text = torch.from_numpy(synth.synth_normal(f"samdec/text/{V}", (1, V, 256), 1.0, 0))
    pe = O.dense_pe(w, SAM_PREFIX + ".prompt_encoder", (64, 64))
    sp, de = O.prompt_encoder_text(w, SAM_PREFIX + ".prompt_encoder", text, (64, 64))
    low, iou = O.mask_decoder(w, SAM_PREFIX + ".mask_decoder", emb, pe, sp, de)
    # multi-view broadcasting: V views become TOKENS (9 = 1 iou + 4 mask + V text), batch V after cross-attn
    assert low.shape == (V, 1, 256, 256) and iou.shape == (V, 1)
    np.testing.assert_allclose(low.numpy(), d["low_res"], atol=1e-5)
    np.testing.assert_allclose(iou.numpy(), d["iou"], atol=1e-5)
    np.testing.assert_allclose(pe.numpy()[..., ::8, ::8], d["dense_pe_sub"], atol=1e-6)
    full = O.postprocess_masks(low, (1024, 1024), (1024, 1024))
    np.testing.assert_allclose(full.numpy()[..., ::16, ::16], d["post_sub"], atol=1e-5)
    assert abs(float(full.double().sum()) - float(d["post_sum"])) < 1e-2
    odd = O.postprocess_masks(low, (1024, 683), (750, 500))
    assert tuple(odd.shape) == tuple(d["post_odd_shape"])
    np.testing.assert_allclose(odd.numpy()[..., ::10, ::10], d["post_odd_sub"], atol=1e-5)


def test_cam_encoders_and_process_embeddings(golden_dir):
    d = _g(golden_dir, "cam_encoders.npz")
    cams = torch.from_numpy(d["cam_params"])
    # normalisation restated in constants.normalize_cam_params (base_contact_dataset.py:37-50)
    np.testing.assert_allclose(cams[1].numpy(), [0.2, 0.875, 0.875, 0.5, 0.65], atol=1e-6)
    emb = torch.from_numpy(synth.synth_normal("cam/seg_emb", (1, 1, 256), 1.0, 0)).repeat(1, 4, 1)
    for k in d.files:
        if k == "cam_params":
            continue
        kind, tt, token = k.split("/")
        w = Wt.synth_weights({**Wt.cam_encoder_spec(kind), **Wt.attention_splitter_spec()})
        cfg = dict(multiview_cam_cond=True, cam_encoder_type=kind, multiview_channels=4, base_token_type=tt,
                   hseg_token_idx=32003, oseg_token_idx=32004)
        r = O.process_embeddings(w, emb.clone(), cams, int(token), cfg)
        np.testing.assert_allclose(r.numpy(), d[k], atol=1e-6, err_msg=k)


def test_sam_image_encoder_small(golden_dir):
    d = _g(golden_dir, "sam_encoder_small.npz")
    c = Wt.SamEncCfg(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,), img_size=480)
    w = Wt.synth_weights(Wt.sam_encoder_spec(c))
    x = torch.from_numpy(synth.synth_normal("samenc/x", (2, 3, 480, 480), 1.0, 0))
    y = O.sam_image_encoder(w, SAM_PREFIX + ".image_encoder", x, 2, 2, (1,))
    np.testing.assert_allclose(y.numpy(), d["out"], atol=1e-5)


def toy_cfg(t):
    return Wt.IvlmCfg(
        llama=Wt.LlamaCfg(hidden=t["hidden"], layers=t["layers"], heads=t["heads"], inter=t["inter"], vocab=t["vocab"]),
        clip=Wt.ClipCfg(hidden=t["clip_hidden"], layers=t["clip_layers"], heads=t["clip_heads"], inter=t["clip_inter"]),
        sam=Wt.SamEncCfg(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,)))


def toy_inputs(d):
    ids = torch.from_numpy(d["input_ids"])
    images_clip = torch.from_numpy(synth.synth_normal("mf/images_clip", (1, 3, 224, 224), 1.0, 0))
    images = torch.from_numpy(synth.synth_normal("mf/images", (1, 4, 3, 1024, 1024), 1.0, 0))[0]
    cams = torch.from_numpy(d["cam_params"])[0]
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    return ids, images_clip, images, cams, tables


def test_model_forward_end_to_end(golden_dir):
    """InteractVLMForCausalLM.model_forward(inference=True) of the reference vs the oracle pipeline."""
    d = _g(golden_dir, "model_forward_toy.npz")
    cfg = toy_cfg(json.loads(str(d["toy"])))
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    ids, images_clip, images, cams, tables = toy_inputs(d)
    o = P.model_forward(w, cfg, images, images_clip, ids, cams, tables)
    np.testing.assert_allclose(o["clip_feat"].numpy(), d["clip_feat"][0], atol=2e-5)
    np.testing.assert_allclose(o["hidden"].numpy(), d["hidden_last"][0], atol=2e-5)
    assert o["hidden"].shape[0] == len(ids) + cfg.img_emb_len  # one -200 expands to 256 rows
    np.testing.assert_allclose(o["low_res"].numpy(), d["low_res"], atol=5e-5)
    pm = o["pred_masks"].numpy()
    np.testing.assert_allclose(pm[..., ::16, ::16], d["pred_masks_sub"], atol=5e-5)
    np.testing.assert_allclose(o["pred_contact"].numpy(), d["pred_contact"], atol=1e-5)
    # [SEG] at id-index 47 selects hidden row 47 - 1 + 255 (InteractVLM.py:331-341)
    rows = O.seg_rows(ids, [32000], 255, model_forward=True)
    assert rows.nonzero().flatten().tolist() == [47 - 1 + 255]


def test_model_forward_oafford_end_to_end(golden_dir):
    """The object-affordance branch of the reference's model_forward(inference=True) ('oafford' sample, 'HM' view type:
    sigmoid on the non-ignored pixels, per-view pixel->point maps, 2048-point lift) vs the oracle pipeline."""
    d = _g(golden_dir, "model_forward_oafford.npz")
    cfg = toy_cfg(json.loads(str(d["toy"])))
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    ids = torch.from_numpy(d["input_ids"])
    images_clip = torch.from_numpy(synth.synth_normal("mf2/images_clip", (1, 3, 224, 224), 1.0, 0))
    images = torch.from_numpy(synth.synth_normal("mf2/images", (1, 4, 3, 1024, 1024), 1.0, 0))[0]
    cams = torch.from_numpy(d["cam_params"])[0]
    pid = synth.synth_point_maps(1, 4, 1024, 1024, 2048, fg=0.3, seed=int(d["point_maps_seed"]))[0]
    valid = np.ones((4, 1024, 1024), bool)
    valid[:, : int(d["ignore_rows"])] = False
    o = P.model_forward_oafford(w, cfg, images, images_clip, ids, cams, pid, valid)
    pm = o["pred_masks"].numpy()
    np.testing.assert_allclose(pm[..., ::16, ::16], d["pred_masks_sub"], atol=5e-5)
    assert abs(float(pm.astype(np.float64).sum()) - float(d["pred_masks_sum"])) < 1e-6 * pm.size
    np.testing.assert_allclose(o["pred_afford"].numpy(), d["pred_afford"], atol=1e-5)
    # what the reference returns for the predictors that do not apply to an 'oafford' sample
    assert d["pred_ocontact"].shape == (1, 0) and float(np.abs(d["pred_human"]).max()) == 0.0


def test_sam_image_encoder_vith_dimensions(golden_dir):
    """The real ViT-H layer dimensions (1280 / 16 heads of 80 / MLP 5120 / 64x64 grid / window 14 + one global block) against
    the reference's ImageEncoderViT (image_encoder.py:110-125), depth 2."""
    d = _g(golden_dir, "sam_encoder_vith_dims.npz")
    c = Wt.SamEncCfg(depth=2, global_attn_indexes=(1,))
    w = Wt.synth_weights(Wt.sam_encoder_spec(c))
    x = torch.from_numpy(synth.synth_normal("samenc_full/x", (1, 3, 1024, 1024), 1.0, 0))
    y = O.sam_image_encoder(w, SAM_PREFIX + ".image_encoder", x, 2, 16, (1,)).numpy()
    np.testing.assert_allclose(y[:, ::4, ::2, ::2], d["out_sub"], atol=2e-5)
    assert abs(float(y.astype(np.float64).sum()) - float(d["out_sum"])) < 1e-5 * y.size


def test_metrics_oracle_vs_reference_golden(golden_dir):
    """oracle/metrics.py against the reference's own get_h_contact_metrics / get_h_geo_metric outputs."""
    import os

    import numpy as np
    import torch

    from oracle import metrics as OM

    d = np.load(os.path.join(golden_dir, "metrics.npz"))
    pred, gt, dist = (torch.from_numpy(d[k]) for k in ("pred", "gt", "dist"))
    fp, fn, per = OM.h_geo_metric(pred, gt, dist)
    np.testing.assert_allclose(per.numpy(), d["geo_per_sample"], rtol=1e-6)
    np.testing.assert_allclose([fp, fn], d["geo_batch"], rtol=1e-6)
    np.testing.assert_allclose(OM.h_contact_metrics(gt, pred).numpy(), d["prf_per_sample"], atol=1e-7)


def _huobj_cfg(t):
    return Wt.IvlmCfg(
        llama=Wt.LlamaCfg(hidden=t["hidden"], layers=t["layers"], heads=t["heads"], inter=t["inter"], vocab=t["vocab"]),
        clip=Wt.ClipCfg(hidden=t["clip_hidden"], layers=t["clip_layers"], heads=t["clip_heads"], inter=t["clip_inter"]),
        sam=Wt.SamEncCfg(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,)), token_type="Gen-Hu-Obj",
        cam_encoder_type="view_index", hseg_token_idx=31999, oseg_token_idx=31998)


def test_model_forward_gen_hu_obj_end_to_end(golden_dir):
    """token_type 'Gen-Hu-Obj' with a [HSEG] answer token and the 'view_index' camera encoder: the AttentionSplitter branch of
    process_embeddings (InteractVLM.py:284-292) inside the reference's whole model_forward(inference=True) vs the oracle."""
    d = _g(golden_dir, "model_forward_huobj.npz")
    cfg = _huobj_cfg(json.loads(str(d["toy"])))
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    ids, images_clip, images, cams, tables = toy_inputs(d)
    assert int(ids[47]) == 31999
    o = P.model_forward(w, cfg, images, images_clip, ids, cams, tables)
    np.testing.assert_allclose(o["low_res"].numpy(), d["low_res"], atol=5e-5)
    np.testing.assert_allclose(o["pred_contact"].numpy(), d["pred_contact"], atol=1e-5)


def _bf16_from_bits(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()


def test_optional_heads_vs_reference_golden(golden_dir):
    """UncertaintyModule / LLaVASAMFusion (components.py:40-153), the reference's own bf16 modules.  bf16 arithmetic is not
    reproducible to the bit across GEMM back ends (accumulation order decides the rounding of each intermediate), so the bound is
    a few bf16 ulps of the value range, not exactness; most elements do agree exactly."""
    d = _g(golden_dir, "optional_heads.npz")
    w = Wt.synth_weights({**Wt.uncertainty_spec(), **Wt.fusion_spec()})
    emb = torch.from_numpy(synth.synth_normal("heads/sam_embeddings", (4, 256, 64, 64), 1.0, 0))
    m = O.uncertainty_head(w, SAM_PREFIX + ".uncertainty", emb)
    ref = _bf16_from_bits(d["uncertainty_map"])
    assert m.shape == ref.shape == (4, 1, 64, 64) and float(ref.min()) > 0
    assert float((m - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float((m == ref).float().mean()) > 0.9
    m1 = O.uncertainty_head(w, SAM_PREFIX + ".uncertainty", emb[:1])
    assert float((m1 - _bf16_from_bits(d["uncertainty_map_V1"])).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    # the resize alone, on the reference's own map: exact with ATen's CPU convention (interpolation weights rounded to bf16),
    # within one ulp of the top binade with fp32 weights (the GPU kernel's convention, the oracle's default); fp32 weights ==
    # F.interpolate in fp32, rounded
    rr = _bf16_from_bits(d["uncertainty_resized"])
    assert torch.equal(O.uncertainty_resize(ref, rr.shape[-2:], lambda_bf16=True), rr)
    r = O.uncertainty_resize(ref, rr.shape[-2:])
    assert float((r - rr).abs().max()) <= 2.0 ** -7 * float(rr.abs().max()) and float((r == rr).float().mean()) > 0.8
    f32 = torch.nn.functional.interpolate(ref, size=tuple(rr.shape[-2:]), mode="bilinear", align_corners=False)
    assert float((r == f32.to(torch.bfloat16).float()).float().mean()) > 0.999

    hw = 16
    sam = torch.from_numpy(synth.synth_normal("heads/fusion_sam", (4, 256, hw, hw), 1.0, 0))
    llava = torch.from_numpy(synth.synth_normal("heads/fusion_llava", (1, 20, 5120), 1.0, 0))
    for key, s_, l_ in (("fusion_V4", sam, llava), ("fusion_V1", sam[:1], llava[:, :7])):
        f = O.sam_fusion(w, SAM_PREFIX + ".fusion", s_, l_)
        fr = _bf16_from_bits(d[key]).reshape(f.shape)
        delta = (fr - s_.to(torch.bfloat16).float())  # what the head adds to the embeddings
        assert float(delta.abs().max()) > 0.05  # (not a no-op)
        assert float((f - fr).abs().max()) <= 2.0 ** -6 * float(fr.abs().max()), key
        assert float((f == fr).float().mean()) > 0.99
