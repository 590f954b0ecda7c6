#!/usr/bin/env python3
"""Per-stage error of the HIP pipeline vs the reference golden taps (toy model_forward fixture)."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from interactvlm_amd import model as M  # noqa: E402
from interactvlm_amd import weights as Wt  # noqa: E402
from test_model_gpu import _toy  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return f"max {float((a-b).abs().max()):.4f} rms {float((a-b).pow(2).mean().sqrt()):.5f} / ref rms {float(b.pow(2).mean().sqrt()):.3f}"


def main():
    dev = torch.device("cuda:0")
    d, cfg, ids, images_clip, images, cams, tables = _toy(os.path.join(REPO, "tests", "golden"))
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
    m.debug_taps = {}
    bf = torch.bfloat16
    out = m.model_forward(images=images.to(bf).to(dev), images_clip=images_clip.to(bf).to(dev), input_ids=ids[None],
                          offset=torch.tensor([0, 1]), masks_list=[torch.zeros(4, 1, 1024, 1024)],
                          label_list=[torch.zeros(1024, 1024)], cam_params=cams, resize_list=[(1024, 1024)],
                          ds_name_list=["hcontact"], mask_paths_list=[None], inference=True)
    t = m.debug_taps
    print("clip_feat ", rel(t["clip_feat"][0], torch.from_numpy(d["clip_feat"][0])))
    print("hidden    ", rel(t["hidden"], torch.from_numpy(d["hidden_last"][0])))
    row = 47 - 1 + 255
    print("seg_fcs   ", rel(t["seg_emb"][0], torch.from_numpy(d["seg_fcs"][0, row])))
    se = t["sam_emb"].view(4, 64, 64, 256).permute(0, 3, 1, 2)[None]
    print("sam_emb   ", rel(se[..., ::4, ::4], torch.from_numpy(d["sam_emb_sub"])))
    print("low_res   ", rel(t["low_res"], torch.from_numpy(d["low_res"])))
    pm = out["pred_masks"][0]
    print("masks     ", rel(pm[..., ::16, ::16], torch.from_numpy(d["pred_masks_sub"])))
    c = out["pred_human_3d_contact"]
    print("contact   ", rel(c, torch.from_numpy(d["pred_contact"])))
    # what if the decoder were fed the reference's exact inputs?  isolates decoder error
    from interactvlm_amd import sam
    ref_emb = torch.from_numpy(d["seg_fcs"][0, row]).to(bf).to(dev)
    e = ref_emb.view(1, 1, 256).repeat(1, 4, 1)
    e = m.process_embeddings(e, cams[0], 32000)
    low2, _ = m.model.visual_model.mask_decoder(t["sam_emb"], e)
    print("low_res | exact seg emb ", rel(low2, torch.from_numpy(d["low_res"])))


if __name__ == "__main__":
    main()
