#!/usr/bin/env python3
"""Wall time of the tail of evaluate() (text_hidden_fcs -> cam conditioning -> SAM mask decoder -> postprocess -> lift) alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import model as M, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    cfg.llama.layers = 2  # the tail does not depend on the depth of the LLM
    w = synthetic.device_weights(cfg, dev, seed=0)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=synthetic.body_lift_tables(dev))
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    emb = m.precompute_visual_embs(im[0])
    out_ids, hidden = m.generate(ic, ids, forced_new_tokens=forced)
    rows = m._seg_rows(out_ids[0].to(dev), extra_false_col=False)

    def tail():
        pm, _ = m._decode_sample(hidden, rows, out_ids[0], cams[0], emb, (S, S), (S, S))
        return m.human_3d_contact_predictor([pm])

    for _ in range(3):
        tail()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        tail()
    torch.cuda.synchronize()
    print(f"tail: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (eager)")


if __name__ == "__main__":
    main()
