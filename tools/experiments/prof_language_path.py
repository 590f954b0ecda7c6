#!/usr/bin/env python3
"""The language path of one evaluate() alone (pre-computed SAM embeddings: no encoder on the side stream), for a kernel trace:
    rocprofv3 --kernel-trace -d /tmp/pl -o l -- python tools/prof_language_path.py; python tools/rocpd_timeline.py <db>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import model as M, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    model = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=synthetic.body_lift_tables(dev))
    del w
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    emb = model.precompute_visual_embs(im[0])
    step = lambda: model.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced, image_embeddings=emb)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        step()
    e.record()
    torch.cuda.synchronize()
    print(f"language path + decoder + lift: {s.elapsed_time(e) / 4:.2f} ms per image")


if __name__ == "__main__":
    main()
