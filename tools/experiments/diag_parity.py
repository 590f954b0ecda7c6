#!/usr/bin/env python3
"""Per-stage error of the HIP pipeline against the fp32 CPU oracle on IDENTICAL bf16-representable weights (the bench's
parity configuration): where the per-vertex contact error comes from.  Test infrastructure (imports oracle/)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from interactvlm_amd import model as M  # noqa: E402
from interactvlm_amd import synth, synthetic  # noqa: E402
from interactvlm_amd import weights as Wt  # noqa: E402
from oracle import pipeline as P  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return (f"max {float((a - b).abs().max()):.3e} rms {float((a - b).pow(2).mean().sqrt()):.3e} | ref max "
            f"{float(b.abs().max()):.3f} rms {float(b.pow(2).mean().sqrt()):.3f}")


def main():
    dev = torch.device("cuda:0")
    cfg = synthetic.config_tiny()
    w = {k: v.to(torch.bfloat16).float() for k, v in Wt.synth_weights(Wt.ivlm_spec(cfg)).items()}
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
    m.debug_taps = {}
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=8)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    full_ids = torch.cat([ids[0], torch.tensor(forced)])
    ref = P.model_forward(w, cfg, im[0].float().cpu(), ic.float().cpu(), full_ids, cams[0], tables)
    for mode in ("evaluate", "model_forward"):
        if mode == "evaluate":
            out = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)
            c = out["pred_contact_3d"]
        else:
            out = m.model_forward(images=im, images_clip=ic, input_ids=full_ids[None], offset=torch.tensor([0, 1]),
                                  masks_list=[torch.zeros(4, 1, 1024, 1024)], label_list=[torch.zeros(1024, 1024)],
                                  cam_params=cams, resize_list=[(1024, 1024)], ds_name_list=["hcontact"],
                                  mask_paths_list=[None], inference=True)
            c = out["pred_human_3d_contact"]
        t = m.debug_taps
        print(f"--- {mode}")
        if "clip_feat" in t:
            print("clip_feat ", rel(t["clip_feat"][0], ref["clip_feat"]))
        n = min(t["hidden"].shape[0], ref["hidden"].shape[0])
        print("hidden    ", rel(t["hidden"][:n], ref["hidden"][:n]))
        print("seg_emb   ", rel(t["seg_emb"], ref["seg_emb"]))
        se = t["sam_emb"].view(4, 64, 64, 256).permute(0, 3, 1, 2)
        print("sam_emb   ", rel(se, ref["sam_emb"]))
        print("low_res   ", rel(t["low_res"], ref["low_res"]))
        print("masks     ", rel(out["pred_masks"][0], ref["pred_masks"]))
        print("contact   ", rel(c, ref["pred_contact"]))
        # decoder alone on the oracle's exact inputs
        emb_o = ref["sam_emb"].permute(0, 2, 3, 1).reshape(4, 4096, 256).to(dev)
        e = ref["seg_emb"].to(dev).view(1, 1, 256).repeat(1, 4, 1)
        e = m.process_embeddings(e, cams[0], cfg.seg_token_idx)
        low2, _ = m.model.visual_model.mask_decoder(emb_o, e)
        print("low_res | oracle seg_emb + oracle sam_emb ", rel(low2, ref["low_res"]))
        low3, _ = m.model.visual_model.mask_decoder(t["sam_emb"], e)
        print("low_res | oracle seg_emb + HIP sam_emb    ", rel(low3, ref["low_res"]))


if __name__ == "__main__":
    main()
