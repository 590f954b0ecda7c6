#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py [--only lift,sam,...]

Every fixture stores the seeded inputs it cannot re-derive plus the reference's outputs.
Weights are never stored: they are ``interactvlm_amd.synth.synth_param(<state-dict key>)``,
poured into the reference's own modules here and into the oracle / HIP path in the tests.
Nothing from the reference's source text is written anywhere — fixtures are data only.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _ref_shims  # noqa: E402
from interactvlm_amd import synth  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------
# F1-F3: lift predictors (model/components.py:195-489)
# ------------------------------------------------------------------------------------------
def gen_lift():
    import torch
    import joblib
    _ref_shims.install()
    import model.components as RC

    V, H, W, NV = 4, 64, 64, 257
    views = RC.HUMAN_VIEW_DICT["4MV-Z_Vitru"]["names"].flatten()
    cwd = os.getcwd()
    for seed in (0, 1, 2):
        vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.4, seed=seed, adversarial=True)
        B = 2 if seed == 0 else 1
        logits = synth.synth_normal(f"lift/logits/{seed}", (B, V, H, W), std=4.0, seed=seed)
        if seed == 1:  # exercise the +-20 clamp (components.py:250)
            logits.reshape(-1)[::97] = 35.0
            logits.reshape(-1)[5::101] = -28.0
        with tempfile.TemporaryDirectory() as td:
            d = os.path.join(td, "data", "hcontact_vitruvian")
            os.makedirs(d)
            np.savez(os.path.join(d, "pixel_to_vertex_map_1024.npz"), **{v: vid[i] for i, v in enumerate(views)})
            np.savez(os.path.join(d, "bary_coords_map_1024.npz"), **{v: bary[i] for i, v in enumerate(views)})
            os.chdir(td)
            try:
                pred = RC.HumanContact3DPredictor("4MV-Z_Vitru", V)
            finally:
                os.chdir(cwd)
        pred.num_vertices = NV  # small-fixture override of the 6890 constant (constants.py:318)
        seg = [torch.from_numpy(logits[b]) for b in range(B)]
        out = pred(seg).numpy()
        _save(f"lift_mesh_soft_s{seed}.npz", logits=logits, vid=vid.astype(np.int32), bary=bary,
              num_vertices=np.int64(NV), expected=out)

        # F2: object-mesh thresholded lift through lift2d_dict.pkl (components.py:392-424)
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "lift2d_dict.pkl")
            joblib.dump({"pixel_to_vertices_map": [vid[i] for i in range(V)],
                         "bary_coords_map": [bary[i] for i in range(V)],
                         "num_vertices": NV}, p)
            om = RC.ObjectMeshContact3DPredictor("4MV-Z_HM", V)
            out_t = om([torch.from_numpy(logits[0])], ds_names=["ocontact"], lift2d_dict_path=p).numpy()
            if seed == 2:  # empty-selection early return (components.py:471-472): all p <= 0.3
                lo = np.full_like(logits[0], -3.0)
                lo[1] = logits[0][1]
                out_e = om([torch.from_numpy(lo)], ds_names=["ocontact"], lift2d_dict_path=p).numpy()
            else:
                lo, out_e = None, None
        extra = {} if lo is None else {"logits_partial": lo, "expected_partial": out_e}
        _save(f"lift_mesh_thresh_s{seed}.npz", logits=logits[0], vid=vid.astype(np.int32), bary=bary,
              num_vertices=np.int64(NV), expected=out_t, **extra)


def gen_lift_points():
    import torch
    _ref_shims.install()
    import model.components as RC

    B, V, H, W, NP = 2, 4, 64, 64, 256
    for seed in (0, 1):
        pid = synth.synth_point_maps(B, V, H, W, NP, fg=0.3, seed=seed)
        probs = synth.synth_uniform(f"lift/probs/{seed}", (B, V, H, W), 0.0, 1.0, seed=seed)
        pc = RC.ObjectPCAfford3DPredictor("4MV-Z_HM", V, num_points=NP)
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for b in range(B):
                row = []
                for v in range(V):
                    mp = os.path.join(td, f"mask_{b}_{v}.png")
                    np.savez(mp.replace("mask", "p2pmap")[:-4] + ".npz", mapping=pid[b, v])
                    row.append(mp)
                paths.append(row)
            out = pc([torch.from_numpy(probs[b]) for b in range(B)], None, paths).numpy()
        # NumPy twin (preprocess_data/utils_obj_pc.py:47-86); its module imports pytorch3d/cv2 at
        # top level for unrelated functions, so those two names are stubbed for the import only.
        import types
        for n, attrs in (("cv2", ()), ("pytorch3d", ()), ("pytorch3d.renderer", (
                "look_at_view_transform", "FoVPerspectiveCameras", "PointsRasterizationSettings",
                "PointsRasterizer", "PointsRenderer", "AlphaCompositor", "NormWeightedCompositor"))):
            if n not in sys.modules:
                m = types.ModuleType(n)
                for a in attrs:
                    setattr(m, a, None)
                sys.modules[n] = m
        from preprocess_data.utils_obj_pc import lift_masks_to_pointcloud
        twin = np.stack([lift_masks_to_pointcloud(list(probs[b]), list(pid[b]), NP) for b in range(B)])
        assert np.array_equal(twin, out), "reference torch predictor and its NumPy twin disagree"
        _save(f"lift_points_s{seed}.npz", probs=probs, pid=pid.astype(np.int32),
              num_points=np.int64(NP), expected=out)


# ------------------------------------------------------------------------------------------
# F4: SAM prompt encoder + mask decoder + postprocess (segment_anything/modeling/*)
# ------------------------------------------------------------------------------------------
def _sub(t, step=16):
    return np.ascontiguousarray(t[..., ::step, ::step])


def gen_sam_decoder():
    import torch
    _ref_shims.install()
    from model.segment_anything.modeling import MaskDecoder, PromptEncoder, TwoWayTransformer
    from model.segment_anything.modeling.sam import Sam
    from interactvlm_amd.weights import SAM_PREFIX

    torch.manual_seed(0)
    pe = PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024), mask_in_chans=16)
    md = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048,
                     num_heads=8), transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    synth.fill_state_dict(pe, 0, SAM_PREFIX + ".prompt_encoder.")
    synth.fill_state_dict(md, 0, SAM_PREFIX + ".mask_decoder.")
    pe.eval(), md.eval()

    class _Enc:  # postprocess_masks only reads image_encoder.img_size
        img_size = 1024

    post = lambda m, i, o: Sam.postprocess_masks(type("S", (), {"image_encoder": _Enc})(), m, i, o)
    for V in (4, 1):
        emb = torch.from_numpy(synth.synth_normal(f"samdec/image_emb/{V}", (V, 256, 64, 64), 1.0, 0))
        text = torch.from_numpy(synth.synth_normal(f"samdec/text/{V}", (1, V, 256), 1.0, 0))
        with torch.no_grad():
            sparse, dense = pe(points=None, boxes=None, masks=None, text_embeds=text)
            low, iou = md(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sparse,
                          dense_prompt_embeddings=dense, multimask_output=False)
            full = post(low, (1024, 1024), (1024, 1024))
            odd = post(low, (1024, 683), (750, 500))
        _save(f"sam_decoder_V{V}.npz", low_res=low.numpy(), iou=iou.numpy(), dense_pe_sub=_sub(pe.get_dense_pe().numpy(), 8),
              post_sub=_sub(full.numpy()), post_sum=np.float64(full.double().sum()),
              post_odd_sub=_sub(odd.numpy(), 10), post_odd_shape=np.array(odd.shape))


# ------------------------------------------------------------------------------------------
# F5: camera-pose encoders + process_embeddings (components.py:491-572, InteractVLM.py:268-294)
# ------------------------------------------------------------------------------------------
def gen_cam():
    import types
    import torch
    _ref_shims.install(full_model=True)
    import model.components as RC
    from model.InteractVLM import InteractVLMForCausalLM
    from interactvlm_amd.constants import HUMAN_VIEW_DICT, normalize_cam_params

    cams = torch.stack([normalize_cam_params(c) for c in HUMAN_VIEW_DICT["4MV-Z_Vitru"]["cam_params"].values()])
    emb = torch.from_numpy(synth.synth_normal("cam/seg_emb", (1, 1, 256), 1.0, 0)).repeat(1, 4, 1)
    out = {"cam_params": cams.numpy()}
    for kind, cls in (("simple", RC.CamPoseEncoder), ("view_index", RC.ViewIndexCamPoseEncoder),
                      ("vi_v1", RC.VIv1CamPoseEncoder)):
        enc = cls() if kind == "simple" else cls(num_views=4)
        synth.fill_state_dict(enc, 0, "cam_pose_encoder.")
        for tt in ("Gen", "Gen-Hu-Obj"):
            ns = types.SimpleNamespace(multiview_cam_cond=True, cam_encoder_type=kind, cam_pose_encoder=enc,
                                       multiview_channels=4, base_token_type=tt, hseg_token_idx=32003,
                                       oseg_token_idx=32004)
            if tt != "Gen":
                ns.attention_splitter = RC.AttentionSplitter()
                synth.fill_state_dict(ns.attention_splitter, 0, "attention_splitter.")
            for token in ((32000,) if tt == "Gen" else (32000, 32003, 32004)):
                with torch.no_grad():
                    r = InteractVLMForCausalLM.process_embeddings(ns, emb.clone(), cams, token)
                out[f"{kind}/{tt}/{token}"] = r.numpy()
    _save("cam_encoders.npz", **out)


# ------------------------------------------------------------------------------------------
# F6: SAM ViT image encoder, reduced width/depth, real head_dim 80 / window 14 / padding / rel-pos
# ------------------------------------------------------------------------------------------
SAM_SMALL = dict(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,), img_size=480)


def gen_sam_encoder():
    from functools import partial
    import torch
    _ref_shims.install()
    from model.segment_anything.modeling import ImageEncoderViT
    from interactvlm_amd.weights import SAM_PREFIX

    c = SAM_SMALL
    enc = ImageEncoderViT(depth=c["depth"], embed_dim=c["embed_dim"], img_size=c["img_size"], mlp_ratio=4,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=c["num_heads"], patch_size=16,
                          qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_attn_indexes"], window_size=14,
                          out_chans=256)
    synth.fill_state_dict(enc, 0, SAM_PREFIX + ".image_encoder.")
    enc.eval()
    x = torch.from_numpy(synth.synth_normal("samenc/x", (2, 3, c["img_size"], c["img_size"]), 1.0, 0))
    with torch.no_grad():
        y = enc(x)
    _save("sam_encoder_small.npz", out=y.numpy())


def gen_sam_encoder_full():
    """F6: the real ViT-H dimensions (1280 wide, 16 heads of 80, MLP 5120, 1024^2 -> 64x64 tokens, one windowed (14) + one
    global block with 127-row rel-pos tables), depth 2, one view.  Output subsampled for the fixture."""
    from functools import partial
    import torch
    _ref_shims.install()
    from model.segment_anything.modeling import ImageEncoderViT
    from interactvlm_amd.weights import SAM_PREFIX

    enc = ImageEncoderViT(depth=2, embed_dim=1280, img_size=1024, mlp_ratio=4,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=16, patch_size=16, qkv_bias=True,
                          use_rel_pos=True, global_attn_indexes=(1,), window_size=14, out_chans=256)
    synth.fill_state_dict(enc, 0, SAM_PREFIX + ".image_encoder.")
    enc.eval()
    x = torch.from_numpy(synth.synth_normal("samenc_full/x", (1, 3, 1024, 1024), 1.0, 0))
    with torch.no_grad():
        y = enc(x)  # [1, 256, 64, 64]
    y = y.numpy()
    _save("sam_encoder_vith_dims.npz", out_sub=y[:, ::4, ::2, ::2].copy(), out_sum=np.float64(y.astype(np.float64).sum()),
          out_abs_mean=np.float64(np.abs(y).mean()))


# ------------------------------------------------------------------------------------------
# F7/F8: the facade — InteractVLMForCausalLM.model_forward(inference=True) on a toy LLaMA/CLIP
# ------------------------------------------------------------------------------------------
TOY = dict(hidden=128, layers=2, heads=4, inter=256, vocab=32003, clip_hidden=64, clip_layers=3, clip_heads=2,
           clip_inter=128, clip_image=224, clip_patch=14)


def gen_model_forward(batch2=False, huobj=False):
    """batch2: second scenario - an oafford sample with the object predictors enabled
    (oC_loss_weight > 0, 'HM' view type: sigmoid on the valid pixels, per-view p2pmap files for the point lift).
    huobj: third scenario - token_type 'Gen-Hu-Obj' with a [HSEG] answer token and the 'view_index' camera encoder: the
    AttentionSplitter branch of process_embeddings (InteractVLM.py:284-292) inside the whole path."""
    import json
    import torch
    _ref_shims.install(full_model=True)
    import model.InteractVLM as RI
    from model.segment_anything.build_sam import _build_sam
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from interactvlm_amd.constants import HUMAN_VIEW_DICT, normalize_cam_params, view_names
    from interactvlm_amd.synth import synth_mesh_tables

    t = TOY
    # reduced SAM encoder (reference's own builder, smaller arguments); img_size stays 1024
    RI.build_sam_vit_h = lambda ckpt=None: _build_sam(encoder_embed_dim=160, encoder_depth=2, encoder_num_heads=2,
                                                     encoder_global_attn_indexes=[1], checkpoint=None)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        clip_dir = os.path.join(td, "clip-toy")
        os.makedirs(clip_dir)
        ccfg = CLIPVisionConfig(hidden_size=t["clip_hidden"], intermediate_size=t["clip_inter"],
                                num_hidden_layers=t["clip_layers"], num_attention_heads=t["clip_heads"],
                                image_size=t["clip_image"], patch_size=t["clip_patch"], hidden_act="quick_gelu")
        ccfg.save_pretrained(clip_dir)
        NV = 6890
        vid, bary = synth_mesh_tables(4, 1024, 1024, NV, fg=0.4, seed=0, patch=8)
        names = view_names(HUMAN_VIEW_DICT["4MV-Z_Vitru"])
        d = os.path.join(td, "data", "hcontact_vitruvian")
        os.makedirs(d)
        np.savez(os.path.join(d, "pixel_to_vertex_map_1024.npz"), **{n: vid[i] for i, n in enumerate(names)})
        np.savez(os.path.join(d, "bary_coords_map_1024.npz"), **{n: bary[i] for i, n in enumerate(names)})
        cfg = RI.LlavaLlamaForCausalLM.config_class(
            hidden_size=t["hidden"], intermediate_size=t["inter"], num_hidden_layers=t["layers"],
            num_attention_heads=t["heads"], num_key_value_heads=t["heads"], vocab_size=t["vocab"], rms_norm_eps=1e-5,
            max_position_embeddings=1024, attn_implementation="eager")
        for k, v in dict(vision_tower=clip_dir, mm_vision_tower=clip_dir, mm_hidden_size=t["clip_hidden"],
                         mm_use_im_start_end=True, mm_vision_select_layer=-2, use_fusion=False, use_uncertainty=False,
                         img_emb_len=255, seg_token_idx=32000, hseg_token_idx=31999 if huobj else None,
                         oseg_token_idx=31998 if huobj else None, token_type="Gen-Hu-Obj" if huobj else "Gen",
                         hC_sam_view_type="4MV-Z_Vitru", oC_sam_view_type="4MV-Z_HM", hC_loss_weight=1.0,
                         oC_loss_weight=1.0 if batch2 else 0.0, multiview_channels=4, multiview_cam_cond=True,
                         cam_encoder_type="view_index" if huobj else "vi_v1",
                         train_mask_decoder=True, out_dim=256).items():
            setattr(cfg, k, v)
        os.chdir(td)
        try:
            torch.manual_seed(0)
            m = RI.InteractVLMForCausalLM(cfg)
            vt = m.get_model().get_vision_tower()
            vt.vision_tower = CLIPVisionModel(ccfg)
            vt.is_loaded = True
        finally:
            os.chdir(cwd)
        m.eval()
        synth.fill_state_dict(m, 0, "")
        # transformers>=5 flattens CLIPVisionModel (no ".vision_model." level); the reference's pinned 4.31 and
        # the openai/clip-vit-large-patch14 checkpoint have it, and so do our keys: re-pour under that name.
        if not any(k.startswith("vision_model.") for k in vt.vision_tower.state_dict()):
            synth.fill_state_dict(vt.vision_tower, 0, "model.vision_tower.vision_tower.vision_model.")
        if batch2:
            # NOTE: two conversation rows ([SEG] tokens) on ONE image is not a runnable scenario of the reference in
            # multiview mode (mask_decoder.py:138 raises a shape error: 2 x V prompt rows vs V dense embeddings), so the
            # object scenario is one oafford sample, like evaluate.py's batch-1 validation loop.
            from interactvlm_amd.constants import OBJS_VIEW_DICT
            from interactvlm_amd.synth import synth_point_maps
            rng = np.random.default_rng(1)
            ids2 = rng.integers(3, 31000, size=52)
            ids2[10], ids2[11], ids2[12] = 32001, -200, 32002
            ids2[44] = 32000
            ids2[51] = 2
            input_ids = torch.from_numpy(ids2)[None]
            images_clip = torch.from_numpy(synth.synth_normal("mf2/images_clip", (1, 3, 224, 224), 1.0, 0))
            images = torch.from_numpy(synth.synth_normal("mf2/images", (1, 4, 3, 1024, 1024), 1.0, 0))
            cams = torch.stack([normalize_cam_params(c) for c in OBJS_VIEW_DICT["4MV-Z_HM"]["cam_params"].values()])[None]
            pid = synth_point_maps(1, 4, 1024, 1024, 2048, fg=0.3, seed=5)[0]  # [V,H,W] int64, -1 = none
            od = os.path.join(td, "obj")
            os.makedirs(od)
            mask_paths = []
            for v in range(4):
                mp = os.path.join(od, f"chair_mask_{v}.png")
                np.savez(mp.replace("mask", "p2pmap")[:-4] + ".npz", mapping=pid[v])
                mask_paths.append(mp)
            gt1 = torch.zeros(4, 1, 1024, 1024)
            gt1[:, :, :100] = -1.0  # IGNORE_LABEL (utils/utils.py:19) band: those pixels keep their logits (no sigmoid)
            with torch.no_grad():
                out = m.model_forward(images=images, images_clip=images_clip, input_ids=input_ids, labels=None,
                                      attention_masks=torch.ones_like(input_ids), offset=torch.tensor([0, 1]),
                                      masks_list=[gt1], label_list=[torch.zeros(1024, 1024)],
                                      gt_contact_3d_list=None, cam_params=cams, resize_list=[(1024, 1024)],
                                      ds_name_list=["oafford_piad"], mask_paths_list=[mask_paths], inference=True)
            pm1 = out["pred_masks"][0].numpy()
            _save("model_forward_oafford.npz", input_ids=ids2, cam_params=cams.numpy(), toy=json.dumps(TOY),
                  point_maps_seed=np.int32(5), ignore_rows=np.int32(100),  # maps: synth_point_maps(1,4,1024,1024,2048,0.3,seed)
                  pred_masks_sub=_sub(pm1), pred_masks_sum=np.float64(pm1.astype(np.float64).sum()),
                  pred_human=out["pred_human_3d_contact"].numpy(),
                  pred_afford=out["pred_object_3d_afford"].numpy(),
                  pred_ocontact=out["pred_object_3d_contact"].numpy())
            return
        # ids: 40 prompt ids with <im_start> <image> <im_end> at 10..12, then a 12-token answer with [SEG]
        rng = np.random.default_rng(0)
        ids = rng.integers(3, 31000, size=52)
        ids[10], ids[11], ids[12] = 32001, -200, 32002
        ids[47] = 31999 if huobj else 32000
        ids[51] = 2
        input_ids = torch.from_numpy(ids)[None]
        images_clip = torch.from_numpy(synth.synth_normal("mf/images_clip", (1, 3, 224, 224), 1.0, 0))
        images = torch.from_numpy(synth.synth_normal("mf/images", (1, 4, 3, 1024, 1024), 1.0, 0))
        cams = torch.stack([normalize_cam_params(c) for c in HUMAN_VIEW_DICT["4MV-Z_Vitru"]["cam_params"].values()])[None]
        # Taps are recorded DURING the one real call: under transformers 5.x a second call of the CLIP tower
        # returns a doubled hidden_states tuple (output-recorder hooks accumulate), so re-calling sub-modules
        # afterwards would tap a different computation than the one that produced pred_masks.
        taps = {}

        def tap(name, pick=lambda o: o):
            def hook(mod, a, o):  # must return None: a returned value would REPLACE the module output
                taps.setdefault(name, pick(o).detach().clone())
            return hook

        gm = m.get_model()
        h1 = gm.mm_projector.register_forward_hook(tap("clip_feat"))
        h2 = gm.norm.register_forward_hook(tap("hidden_last"))
        h3 = gm.text_hidden_fcs[0].register_forward_hook(tap("fcs"))
        h4 = gm.visual_model.image_encoder.register_forward_hook(tap("sam_emb"))
        h5 = gm.visual_model.mask_decoder.register_forward_hook(tap("low_res", lambda o: o[0]))
        with torch.no_grad():
            out = m.model_forward(images=images, images_clip=images_clip, input_ids=input_ids, labels=None,
                                  attention_masks=torch.ones_like(input_ids), offset=torch.tensor([0, 1]),
                                  masks_list=[torch.zeros(4, 1, 1024, 1024)], label_list=[torch.zeros(1024, 1024)],
                                  gt_contact_3d_list=None, cam_params=cams, resize_list=[(1024, 1024)],
                                  ds_name_list=["hcontact"], mask_paths_list=[None], inference=True)
        for h in (h1, h2, h3, h4, h5):
            h.remove()
        clip_feat, hid, img_emb = taps["clip_feat"], taps["hidden_last"], taps["sam_emb"]
    pm = out["pred_masks"][0].numpy()
    _save("model_forward_huobj.npz" if huobj else "model_forward_toy.npz", input_ids=ids, cam_params=cams.numpy(), toy=json.dumps(TOY),
          clip_feat=clip_feat.numpy(), hidden_last=hid.numpy(), sam_emb_sub=_sub(img_emb.numpy(), 4),
          seg_fcs=taps["fcs"].numpy(), low_res=taps["low_res"].numpy(),
          pred_masks_sub=_sub(pm), pred_masks_sum=np.float64(pm.astype(np.float64).sum()),
          pred_contact=out["pred_human_3d_contact"].numpy())


# ------------------------------------------------------------------------------------------
# a17: caller-side preprocessing - the reference's ResizeLongestSide (segment_anything/utils/transforms.py:17-34,102-113) and
# HF CLIPImageProcessor with the openai/clip-vit-large-patch14 settings (run_demo.py:170,333-339), non-identity sizes
# ------------------------------------------------------------------------------------------
PREPROCESS_SIZES = [(600, 1500), (480, 640), (1024, 1024), (333, 500), (1500, 600)]


def gen_preprocess():
    import torch
    from PIL import Image
    _ref_shims.install()
    # torchvision is absent: ResizeLongestSide.apply_image only needs two of its functions; on a PIL image they are
    # Image.fromarray and Image.resize(..., BILINEAR) (torchvision.transforms.functional.resize's PIL branch)
    tvf = sys.modules["torchvision.transforms.functional"]
    tvf.to_pil_image = lambda a: Image.fromarray(a)
    tvf.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)
    import importlib
    T = importlib.import_module("model.segment_anything.utils.transforms")
    T.resize, T.to_pil_image = tvf.resize, tvf.to_pil_image
    from transformers import CLIPImageProcessorPil
    proc = CLIPImageProcessorPil(do_resize=True, size={"shortest_edge": 224}, resample=Image.BICUBIC, do_center_crop=True,
                                 crop_size={"height": 224, "width": 224}, do_rescale=True, rescale_factor=1 / 255,
                                 do_normalize=True, image_mean=[0.48145466, 0.4578275, 0.40821073],
                                 image_std=[0.26862954, 0.26130258, 0.27577711], do_convert_rgb=True)
    tr = T.ResizeLongestSide(1024)
    mean = torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
    out = {"sizes": np.array(PREPROCESS_SIZES)}
    for h, w in PREPROCESS_SIZES:
        img = np.random.default_rng(h * 10000 + w).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        r = tr.apply_image(img)  # uint8 [h', w', 3]
        x = torch.from_numpy(r).permute(2, 0, 1).contiguous().float()
        x = (x - mean) / std  # run_demo.py:65-79 preprocess(): normalise, then zero-pad to 1024^2
        x = torch.nn.functional.pad(x, (0, 1024 - x.shape[2], 0, 1024 - x.shape[1]))
        out[f"sam/{h}x{w}/resize"] = np.array(r.shape[:2])
        out[f"sam/{h}x{w}/sub"] = x[:, ::8, ::8].numpy()
        out[f"sam/{h}x{w}/sum"] = np.float64(x.double().sum())
        pv = proc(Image.fromarray(img), return_tensors="pt")["pixel_values"][0]
        out[f"clip/{h}x{w}/sub"] = pv[:, ::2, ::2].numpy()
        out[f"clip/{h}x{w}/sum"] = np.float64(pv.double().sum())
    _save("preprocess.npz", **out)


# ------------------------------------------------------------------------------------------
# f3: the state-dict key inventory of the reference's own module tree (what merge_lora_weights_and_save_hf_model.py:152-161
# writes into a released checkpoint: everything except vision_tower.*), for the configurations the loader must accept
# ------------------------------------------------------------------------------------------
def gen_state_keys():
    import json
    import torch
    _ref_shims.install(full_model=True)
    import model.InteractVLM as RI
    from model.segment_anything.build_sam import _build_sam
    from transformers import CLIPVisionConfig
    from interactvlm_amd.constants import HUMAN_VIEW_DICT, view_names
    from interactvlm_amd.synth import synth_mesh_tables

    t = TOY
    RI.build_sam_vit_h = lambda ckpt=None: _build_sam(encoder_embed_dim=160, encoder_depth=2, encoder_num_heads=2,
                                                     encoder_global_attn_indexes=[1], checkpoint=None)
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        clip_dir = os.path.join(td, "clip-toy")
        os.makedirs(clip_dir)
        CLIPVisionConfig(hidden_size=t["clip_hidden"], intermediate_size=t["clip_inter"], num_hidden_layers=t["clip_layers"],
                         num_attention_heads=t["clip_heads"], image_size=t["clip_image"], patch_size=t["clip_patch"],
                         hidden_act="quick_gelu").save_pretrained(clip_dir)
        vid, bary = synth_mesh_tables(4, 64, 64, 6890, fg=0.4, seed=0, patch=8)
        names = view_names(HUMAN_VIEW_DICT["4MV-Z_Vitru"])
        d = os.path.join(td, "data", "hcontact_vitruvian")
        os.makedirs(d)
        np.savez(os.path.join(d, "pixel_to_vertex_map_1024.npz"), **{n: vid[i] for i, n in enumerate(names)})
        np.savez(os.path.join(d, "bary_coords_map_1024.npz"), **{n: bary[i] for i, n in enumerate(names)})
        for tag, token_type, cam, oc in (("Gen/vi_v1", "Gen", "vi_v1", 0.0), ("Gen/simple", "Gen", "simple", 0.0),
                                         ("Gen-Hu-Obj/view_index", "Gen-Hu-Obj", "view_index", 1.0),
                                         ("Gen-Hu-Obj-DifDe/vi_v1", "Gen-Hu-Obj-DifDe", "vi_v1", 1.0),
                                         ("Gen/vi_v1/heads", "Gen", "vi_v1", 0.0)):  # .../heads: use_fusion + use_uncertainty
            heads = tag.endswith("/heads")
            cfg = RI.LlavaLlamaForCausalLM.config_class(
                hidden_size=t["hidden"], intermediate_size=t["inter"], num_hidden_layers=t["layers"],
                num_attention_heads=t["heads"], num_key_value_heads=t["heads"], vocab_size=t["vocab"], rms_norm_eps=1e-5,
                max_position_embeddings=1024, attn_implementation="eager")
            for k, v in dict(vision_tower=clip_dir, mm_vision_tower=clip_dir, mm_hidden_size=t["clip_hidden"],
                             mm_use_im_start_end=True, mm_vision_select_layer=-2, use_fusion=heads, use_uncertainty=heads,
                             img_emb_len=255, seg_token_idx=32000, hseg_token_idx=32003, oseg_token_idx=32004,
                             token_type=token_type, hC_sam_view_type="4MV-Z_Vitru", oC_sam_view_type="4MV-Z_HM",
                             hC_loss_weight=1.0, oC_loss_weight=oc, multiview_channels=4, multiview_cam_cond=True,
                             cam_encoder_type=cam, train_mask_decoder=True, out_dim=256).items():
                setattr(cfg, k, v)
            os.chdir(td)
            try:
                m = RI.InteractVLMForCausalLM(cfg)
            finally:
                os.chdir(cwd)
            out[tag] = {k: list(v.shape) for k, v in m.state_dict().items() if "vision_tower" not in k}
            del m
    path = os.path.join(HERE, "state_dict_keys.json")
    with open(path, "w") as f:
        json.dump({"toy": TOY, "configs": out}, f, indent=0, sort_keys=True)
    print(f"  wrote state_dict_keys.json ({os.path.getsize(path) / 1024:.1f} KiB, {', '.join(f'{k}: {len(v)}' for k, v in out.items())})")


# ------------------------------------------------------------------------------------------
# metrics right after the path (utils/eval_utils.py:63-151): get_h_contact_metrics, get_h_geo_metric
# ------------------------------------------------------------------------------------------
def gen_metrics():
    import torch

    _ref_shims.install()
    n = 211  # synthetic "mesh" size: the distance matrix is an input (the real one is a 190 MB data file we do not have)
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    dist = np.linalg.norm(pts[:, None] - pts[None], axis=-1).astype(np.float32)
    dist = dist + 0.05 * rng.random((n, n)).astype(np.float32) * (1 - np.eye(n, dtype=np.float32))  # NOT symmetric
    real_load = np.load

    def fake_load(path, *a, **k):  # eval_utils.py:15 loads the geodesic matrix at import time
        return dist if "geodesic" in str(path) else real_load(path, *a, **k)

    np.load = fake_load
    try:
        import utils.eval_utils as E
    finally:
        np.load = real_load
    E.DIST_MATRIX = torch.from_numpy(dist)
    B = 5
    pred = rng.random((B, n)).astype(np.float32)
    gt = (rng.random((B, n)) < 0.3).astype(np.float32)
    pred[1] = 0.1          # no vertex predicted in contact: every row is used (eval_utils.py:141)
    gt[2] = 0.0            # no ground-truth contact: every column is used (:140)
    pred[3, :5] = 0.5      # the >= 0.5 boundary
    gt[4] = gt[4] * 0.7    # gt values other than exactly 1 are not contact columns for the geodesic metric, but are > 0 for F1
    gt[4, :3] = 1.0
    pt, gtt = torch.from_numpy(pred), torch.from_numpy(gt)
    geo = [E.get_h_geo_metric(pt[b: b + 1], gtt[b: b + 1]) for b in range(B)]
    geo_all = E.get_h_geo_metric(pt, gtt)
    prf = [E.get_h_contact_metrics(gtt[b: b + 1], pt[b: b + 1]) for b in range(B)]
    prf_o = [E.get_o_contact_metrics(gtt[b: b + 1], pt[b: b + 1]) for b in range(B)]
    _save("metrics.npz", dist=dist, pred=pred, gt=gt, geo_per_sample=np.asarray(geo, np.float64),
          geo_batch=np.asarray(geo_all, np.float64), prf_per_sample=np.asarray(prf, np.float64),
          prf_o_per_sample=np.asarray(prf_o, np.float64))


# ------------------------------------------------------------------------------------------
# optional heads of ModifiedSAM (model/components.py:40-153; off in every released configuration): the reference's own modules
# in bf16 (both cast their inputs to bf16, so they only run inside the bf16 model), inputs re-derivable from their synth keys
# ------------------------------------------------------------------------------------------
HEADS_UNC_SIZE = (187, 250)     # evaluate()'s original_size for the uncertainty map (InteractVLM.py:614-617)
HEADS_FUSION_HW = 16            # spatial side of the fusion fixture (the module takes H, W from its input)
HEADS_FUSION_L = 20             # LLaVA positions (the module's view(batch, -1, ...) splits them over the 4 views)


def _bf16_bits(t):
    import torch
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def gen_optional_heads():
    import torch
    import torch.nn.functional as F
    _ref_shims.install()
    import model.components as RC

    out = {}
    unc = RC.UncertaintyModule()
    synth.fill_state_dict(unc, 0, "model.visual_model.uncertainty.")
    unc.bfloat16()
    emb = torch.from_numpy(synth.synth_normal("heads/sam_embeddings", (4, 256, 64, 64), 1.0, 0))
    with torch.no_grad():
        m = unc(emb)  # [4,1,64,64] bf16
        r = F.interpolate(m, size=HEADS_UNC_SIZE, mode="bilinear", align_corners=False)
        m1 = unc(emb[:1])
    assert m.dtype == torch.bfloat16 and r.dtype == torch.bfloat16
    out["uncertainty_map"] = _bf16_bits(m)
    out["uncertainty_resized"] = _bf16_bits(r)
    out["uncertainty_map_V1"] = _bf16_bits(m1)

    fus = RC.LLaVASAMFusion()
    synth.fill_state_dict(fus, 0, "model.visual_model.fusion.")
    fus.bfloat16()
    hw = HEADS_FUSION_HW
    sam = torch.from_numpy(synth.synth_normal("heads/fusion_sam", (4, 256, hw, hw), 1.0, 0))
    llava = torch.from_numpy(synth.synth_normal("heads/fusion_llava", (1, HEADS_FUSION_L, 5120), 1.0, 0))
    with torch.no_grad():
        f4 = fus(sam, llava)                       # the 4 views attend to consecutive quarters of the LLaVA positions
        f1 = fus(sam[:1], llava[:, :7])            # one view: all positions
    out["fusion_V4"] = _bf16_bits(f4)
    out["fusion_V1"] = _bf16_bits(f1)
    _save("optional_heads.npz", **out)


# ------------------------------------------------------------------------------------------
# Prompt construction: conv_templates["llava_v1"] (model/llava/conversation.py:355-365) and tokenizer_image_token
# (model/llava/mm_utils.py:31-56) as run_demo.py:313-324 drives them, on the demo's own three questions
# ------------------------------------------------------------------------------------------
def gen_prompt():
    import ast
    import json

    _ref_shims.install(full_model=True)  # (model.llava's package import pulls the LLaVA model classes in)
    from model.llava import conversation as conversation_lib
    from model.llava.mm_utils import tokenizer_image_token
    from utils.utils import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX

    from stub_tokenizer import StubTokenizer

    # the three BASE_PROMPT lists are literals inside run_demo.main(): read the VALUES out of its syntax tree (data, not code)
    tree = ast.parse(open(os.path.join(_ref_shims.REFERENCE_ROOT, "run_demo.py")).read())
    base = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "BASE_PROMPT" for t in node.targets):
            base.append((node.lineno, ast.literal_eval(node.value)[0]))
    base.sort()
    assert len(base) == 3, base
    names = ("oafford", "h2dcontact", "hcontact")  # (their order in the file: run_demo.py:217, 254, 282)
    cases = []
    for (lineno, question), name in zip(base, names):
        q = question.format(class_name="chair", object="chair")
        for mm in (True, False):
            conv = conversation_lib.conv_templates["llava_v1"].copy()  # run_demo.py:313-324
            conv.messages = []
            prompt = DEFAULT_IMAGE_TOKEN + "\n" + q
            if mm:
                prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN)
            conv.append_message(conv.roles[0], prompt)
            conv.append_message(conv.roles[1], "")
            text = conv.get_prompt()
            ids = {("bos" if bos else "nobos"): tokenizer_image_token(text, StubTokenizer(bos), return_tensors="pt").tolist()
                   for bos in (True, False)}
            cases.append({"kind": name, "line": lineno, "template": question, "question": q, "use_mm_start_end": mm, "prompt": text,
                          "ids": ids})
    # the splice logic on its own: several placeholders, one at the very start / end, none at all
    splice = []
    for text in ("a b <image> c d <image> e", "<image> a", "a <image>", "a b c", "<image>"):
        splice.append({"text": text, "ids": {("bos" if bos else "nobos"): tokenizer_image_token(text, StubTokenizer(bos))
                                             for bos in (True, False)}})
    out = {"image_token_index": IMAGE_TOKEN_INDEX, "tokens": [DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN],
           "cases": cases, "splice": splice}
    path = os.path.join(HERE, "prompts.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"  wrote prompts.json ({len(cases)} prompts, {len(splice)} splice cases)")


GENERATORS = {"optional_heads": gen_optional_heads, "lift": gen_lift, "lift_points": gen_lift_points, "sam_decoder": gen_sam_decoder, "cam": gen_cam,
              "sam_encoder": gen_sam_encoder, "sam_encoder_full": gen_sam_encoder_full, "model_forward": gen_model_forward,
              "model_forward_oafford": lambda: gen_model_forward(batch2=True),
              "model_forward_huobj": lambda: gen_model_forward(huobj=True), "metrics": gen_metrics,
              "state_keys": gen_state_keys, "preprocess": gen_preprocess, "prompt": gen_prompt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    for name, fn in GENERATORS.items():
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()


if __name__ == "__main__":
    main()
