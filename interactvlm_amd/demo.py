"""The reference's demo driver for one sample (run_demo.py:273-470), as a library: prompt assembly, image preprocessing,
``evaluate`` and the ``*_vertices.npz`` outputs.  Rendering the object views is ``render.object_lift_tables``.

Everything that touches pixels or vertices runs on the GPU (preprocess.py, model.py, ops.SparseRows); the tokenizer is
whatever the caller loaded (any object with ``__call__(text).input_ids`` and ``bos_token_id``, like the HF tokenizer at
run_demo.py:87-118) - no tokenizer ships with the reference.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops, preprocess
from .constants import HUMAN_VIEW_DICT, IMAGE_TOKEN_INDEX, OBJS_VIEW_DICT, normalize_cam_params

DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<image>", "<im_start>", "<im_end>"  # utils/utils.py:12-17
HCONTACT_PROMPT = "Which body parts are in contact with the {object}? Segment these contact areas."        # run_demo.py:282
H2DCONTACT_PROMPT = "Segment the area on the human's body that is in direct contact with the {object} in this image."  # run_demo.py:254
# run_demo.py:217 - the literal the reference sends INCLUDES the stray quotes and the trailing comma of its list entry
OAFFORD_PROMPT = ('"What type of affordance does the human-object interaction suggest? Then, segment the area on the {class_name} where '
                  'the human is making contact.",')
# model/llava/conversation.py:355-365 (conv_llava_v1: SeparatorStyle.TWO, sep ' ', sep2 '</s>')
_V1_SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the human's questions.")


def build_prompt(question: str, use_mm_start_end: bool = True) -> str:
    """run_demo.py:313-324 with conv_type 'llava_v1': one USER turn holding the image token, empty ASSISTANT turn."""
    user = DEFAULT_IMAGE_TOKEN + "\n" + question
    if use_mm_start_end:
        user = user.replace(DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN)
    return _V1_SYSTEM + " " + "USER: " + user + " " + "ASSISTANT:"


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> torch.Tensor:
    """model/llava/mm_utils.py:16-44: tokenise the text around every '<image>' and splice the placeholder id in between
    (a leading BOS is kept once)."""
    chunks = [list(tokenizer(c).input_ids) for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids: List[int] = []
    offset = 0
    if chunks and chunks[0] and chunks[0][0] == getattr(tokenizer, "bos_token_id", None):
        offset = 1
        ids.append(chunks[0][0])
    for i, c in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)  # the separator [image] * (offset + 1) minus its first `offset` entries
        ids.extend(c[offset:])
    return torch.tensor(ids, dtype=torch.long)


def cam_params_for(contact_type: str, view_type: str) -> torch.Tensor:
    """[1, V, 5] normalised camera parameters in view order (run_demo.py:276-278, 205-210)."""
    table = HUMAN_VIEW_DICT if "hcontact" in contact_type else OBJS_VIEW_DICT
    cams = table[view_type]["cam_params"]
    return torch.stack([normalize_cam_params(c) for c in cams.values()])[None]


@torch.no_grad()
def generate_sam_inp_objs(verts, faces, out_dir: str, view_type: str = "4MV-Z_HM_BM", colored: bool = True,
                          image_size=(1024, 1024)):
    """generate_sam_inp_objs (utils/demo_utils.py:171-257) on the GPU: normalise the object mesh, rasterise + Phong-shade
    the four object views, write ``lift2d_dict.pkl`` (the file ObjectMeshContact3DPredictor reads) into ``out_dir``.
    verts f32 [Nv,3] / faces i32 [Nf,3] GPU tensors -> (sam_views: 4 uint8 [H,W,3] arrays, lift2d_dict_path)."""
    from . import render

    imgs, vid, bary, nv = render.object_renders(verts, faces, view_type, colored=colored, image_size=image_size)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "lift2d_dict.pkl")
    render.save_lift2d_dict(path, vid, bary, nv)
    return [im.cpu().numpy() for im in imgs], path


def run_sample(model, image_rgb: np.ndarray, sam_views: Sequence[np.ndarray], input_ids: torch.Tensor, contact_type="hcontact",
               out_dir: Optional[str] = None, name: str = "sample", lift2d_dict_path: Optional[str] = None,
               smpl_to_smplx: Optional[ops.SparseRows] = None, max_new_tokens: int = 512, forced_new_tokens=None,
               image_embeddings=None):
    """One demo sample (run_demo.py:325-392, 436-456).

    image_rgb uint8 [H,W,3] (the photo, CLIP input); sam_views: V uint8 [h,w,3] renders (SAM inputs); input_ids [L] with one
    -200.  Writes ``{name}_hcontact_vertices.npz`` (pred_contact_3d_smplh [+ pred_contact_3d_smplx through the 3-nnz/row
    SpMV]) or ``{name}_oafford_vertices.npz`` (pred_contact_3d) like the reference, returns evaluate()'s dict."""
    dev = model.device
    image_clip = preprocess.clip_preprocess(image_rgb, dev)[None]
    views, resize = [], None
    for v in sam_views:
        t, hw = preprocess.sam_preprocess(np.asarray(v), dev, model.config.sam.img_size)
        views.append(t)
        resize = resize or hw
    sam_multiview = torch.stack(views)[None]
    view_type = model.config.hC_sam_view_type if "hcontact" in contact_type else model.config.oC_sam_view_type
    cams = cam_params_for(contact_type, view_type)
    out = model.evaluate(image_clip, sam_multiview, input_ids[None], cams, [resize], [resize], lift2d_dict_path=lift2d_dict_path,
                         contact_type=contact_type, max_new_tokens=max_new_tokens, forced_new_tokens=forced_new_tokens,
                         image_embeddings=image_embeddings)
    pc = out["pred_contact_3d"]
    if out_dir is not None and pc is not None:
        os.makedirs(out_dir, exist_ok=True)
        if contact_type == "hcontact":
            arrays = {"pred_contact_3d_smplh": pc.float().cpu().numpy()}
            if smpl_to_smplx is not None:  # utils/utils.py:428-443 convert_contacts (dense bmm there, SpMV here)
                arrays["pred_contact_3d_smplx"] = smpl_to_smplx.matvec(pc.float().contiguous()).squeeze().cpu().numpy()
            np.savez(os.path.join(out_dir, f"{name}_hcontact_vertices.npz"), **arrays)
        else:
            np.savez(os.path.join(out_dir, f"{name}_oafford_vertices.npz"), pred_contact_3d=pc.float().cpu().numpy())
    return out
