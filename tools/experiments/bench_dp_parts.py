#!/usr/bin/env python3
"""Where the time of one 8-image evaluate_batch call (dp64's unit) goes: SAM encoder alone (32 views in one pass and 8 x 4),
the language path alone (cached embeddings), both together."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import model as M
    from interactvlm_amd import synthetic

    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    vid, bary = synthetic.body_lift_tables(dev)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=(vid, bary))
    del w
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    B, S = 8, cfg.sam.img_size
    ic, im = synthetic.images(cfg, dev, seed=5, batch=B)
    enc = m.model.visual_model.image_encoder

    def t(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print(f"SAM encoder, 32 views in one pass: {t(lambda: enc(im.reshape(B * 4, 3, S, S))):.1f} ms", flush=True)
    print(f"SAM encoder, 8 passes of 4 views:  {t(lambda: [enc(im[b]) for b in range(B)]):.1f} ms", flush=True)
    emb = [enc(im[b]) for b in range(B)]
    args = ([ids[0]] * B, [cams[0]] * B, [(S, S)] * B, [(S, S)] * B)
    print(f"language path + decoder + lift (cached embeddings): "
          f"{t(lambda: m.evaluate_batch(ic, None, *args, forced_new_tokens=forced, image_embeddings=emb)):.1f} ms", flush=True)
    print(f"generate_batch alone: {t(lambda: m.generate_batch(ic, [ids[0]] * B, forced_new_tokens=forced)):.1f} ms", flush=True)
    print(f"evaluate_batch (overlapped): {t(lambda: m.evaluate_batch(ic, im, *args, forced_new_tokens=forced)):.1f} ms", flush=True)
    m.overlap_sam_encoder = False
    print(f"evaluate_batch (serial):     {t(lambda: m.evaluate_batch(ic, im, *args, forced_new_tokens=forced)):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
