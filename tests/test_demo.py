"""run_demo-equivalent plumbing: prompt assembly, image-token splicing, and one sample end to
end on the GPU with the reference's output files."""
import os

import numpy as np
import pytest
import torch

from interactvlm_amd import demo


class _Tok:
    """minimal stand-in for the HF tokenizer interface used by tokenizer_image_token: BOS + one id per word"""
    bos_token_id = 1

    def __init__(self, bos=True):
        self.bos = bos
        self.vocab = {}

    def __call__(self, text):
        ids = [self.vocab.setdefault(w, 10 + len(self.vocab)) for w in text.split()]

        class R:
            input_ids = ([1] if self.bos else []) + ids
        return R


def test_prompt_and_image_token_splicing():
    p = demo.build_prompt(demo.HCONTACT_PROMPT.format(object="chair"))
    assert p.startswith("A chat between a curious human") and p.endswith("Segment these contact areas. ASSISTANT:")
    assert "USER: <im_start><image><im_end>\nWhich body parts" in p
    tok = _Tok(bos=True)
    ids = demo.tokenizer_image_token("a b <image> c d <image> e", tok).tolist()
    a, b, c, d, e = (tok.vocab[w] for w in "abcde")
    assert ids == [1, a, b, -200, c, d, -200, e]          # one BOS, one placeholder per <image> (mm_utils.py:16-44)
    tok2 = _Tok(bos=False)
    ids2 = demo.tokenizer_image_token("a <image> b", tok2).tolist()
    assert ids2 == [tok2.vocab["a"], -200, tok2.vocab["b"]]
    cams = demo.cam_params_for("hcontact", "4MV-Z_Vitru")
    assert cams.shape == (1, 4, 5) and abs(float(cams[0, 0, 0]) - 0.2) < 1e-6   # d / 10 (base_contact_dataset.py:37-50)


def test_prompt_construction_equals_the_reference(golden_dir):
    """``demo.build_prompt`` / ``demo.tokenizer_image_token`` against what the REFERENCE's own
    ``conv_templates["llava_v1"]`` (model/llava/conversation.py:355-365) and ``tokenizer_image_token`` (model/llava/mm_utils.py:31-56)
    produce when driven as run_demo.py:313-324 drives them - on the demo's three questions (run_demo.py:217, 254, 282; the texts
    themselves were read out of the reference), with use_mm_start_end on / off, with and without a BOS id, plus the splice logic on
    texts with several / leading / trailing / no image placeholders.  Fixture: tests/golden/prompts.json (make_golden.py gen_prompt)."""
    import json
    import sys

    sys.path.insert(0, golden_dir)
    from stub_tokenizer import StubTokenizer

    with open(os.path.join(golden_dir, "prompts.json")) as f:
        g = json.load(f)
    assert g["image_token_index"] == demo.IMAGE_TOKEN_INDEX
    assert g["tokens"] == [demo.DEFAULT_IMAGE_TOKEN, demo.DEFAULT_IM_START_TOKEN, demo.DEFAULT_IM_END_TOKEN]
    templates = {"oafford": demo.OAFFORD_PROMPT, "h2dcontact": demo.H2DCONTACT_PROMPT, "hcontact": demo.HCONTACT_PROMPT}
    assert len(g["cases"]) == 6
    for c in g["cases"]:
        assert templates[c["kind"]] == c["template"]  # the question texts are the reference's, character for character
        q = templates[c["kind"]].format(class_name="chair", object="chair")
        assert q == c["question"]
        p = demo.build_prompt(q, use_mm_start_end=c["use_mm_start_end"])
        assert p == c["prompt"]
        for key, bos in (("bos", True), ("nobos", False)):
            assert demo.tokenizer_image_token(p, StubTokenizer(bos)).tolist() == c["ids"][key], (c["kind"], key)
            assert c["ids"][key].count(-200) == 1
    for c in g["splice"]:
        for key, bos in (("bos", True), ("nobos", False)):
            assert demo.tokenizer_image_token(c["text"], StubTokenizer(bos)).tolist() == c["ids"][key], (c["text"], key)


@pytest.mark.gpu
def test_run_sample_writes_reference_outputs(hip_lib, cuda, tmp_path):
    from interactvlm_amd import model as M
    from interactvlm_amd import ops, synth, synthetic
    from interactvlm_amd import weights as Wt

    cfg = synthetic.config_tiny()
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, cuda, lift_tables=tables)
    rng = np.random.default_rng(0)
    photo = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)                    # any size: CLIP resize + crop
    renders = [rng.integers(0, 256, size=(1024, 1024, 3), dtype=np.uint8) for _ in range(4)]
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=6)
    mapping = torch.zeros(10475, 6890)
    r = torch.arange(10475)
    for k in range(3):  # 3 non-zeros per row like the SMPL -> SMPL-X barycentric transfer
        mapping[r, (r * 7 + k * 13) % 6890] += [0.5, 0.3, 0.2][k]
    out = demo.run_sample(m, photo, renders, ids[0], "hcontact", out_dir=str(tmp_path), name="chair__img",
                          smpl_to_smplx=ops.SparseRows(mapping, cuda), forced_new_tokens=forced)
    z = np.load(os.path.join(tmp_path, "chair__img_hcontact_vertices.npz"))
    assert z["pred_contact_3d_smplh"].shape == (1, 6890) and z["pred_contact_3d_smplx"].shape == (10475,)
    pc = out["pred_contact_3d"].float().cpu()
    assert np.array_equal(z["pred_contact_3d_smplh"], pc.numpy())
    np.testing.assert_allclose(z["pred_contact_3d_smplx"], (mapping @ pc[0]).numpy(), atol=1e-5)
    assert out["pred_masks"][0].shape == (4, 1024, 1024)


@pytest.mark.gpu
def test_object_sample_from_a_mesh_end_to_end(hip_lib, cuda, tmp_path):
    """run_demo's object branch from nothing but a mesh (utils/demo_utils.py:171-257 then run_demo.py:325-392): normalise,
    rasterise and Phong-shade the four object views on the GPU, write lift2d_dict.pkl, run evaluate('ocontact') on the
    renders and write ``*_oafford_vertices.npz`` - one contact value per mesh vertex."""
    from interactvlm_amd import model as M
    from interactvlm_amd import synth, synthetic
    from interactvlm_amd import weights as Wt
    from oracle import raster as R

    v, f = R.icosphere(3)
    v = (v * np.array([1.0, 0.5, 0.7], np.float32)).astype(np.float32)
    vt, ft = torch.from_numpy(v).to(cuda), torch.from_numpy(f.astype(np.int32)).to(cuda)
    views, path = demo.generate_sam_inp_objs(vt, ft, str(tmp_path / "sam_inp_objs"))
    assert len(views) == 4 and views[0].shape == (1024, 1024, 3) and views[0].dtype == np.uint8 and os.path.exists(path)
    assert (views[0][0, 0] == 255).all() and len(np.unique(views[0].reshape(-1, 3), axis=0)) > 50  # white bg, shaded object
    cfg = synthetic.config_tiny()
    cfg.oC_loss_weight, cfg.oC_sam_view_type = 1.0, "4MV-Z_HM_BM"
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, cuda, lift_tables=tables)
    photo = np.random.default_rng(1).integers(0, 256, size=(600, 400, 3), dtype=np.uint8)
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=6)
    out = demo.run_sample(m, photo, views, ids[0], "ocontact", out_dir=str(tmp_path), name="mug__img", lift2d_dict_path=path,
                          forced_new_tokens=forced)
    z = np.load(os.path.join(tmp_path, "mug__img_oafford_vertices.npz"))
    assert z["pred_contact_3d"].shape == (1, v.shape[0]) and np.isfinite(z["pred_contact_3d"]).all()
    assert np.array_equal(z["pred_contact_3d"], out["pred_contact_3d"].float().cpu().numpy())
