"""Where the time of evaluate_batch goes (7B shapes, synthetic weights): generate_batch alone for B = 1..8 (CLIP + prefill +
batched decode), the SAM encoder alone, and the whole call.  Usage: python tools/bench_batch.py [--model 7b]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from interactvlm_amd import model as M
    from interactvlm_amd import synthetic

    dev = torch.device("cuda:0")
    cfg = {"7b": synthetic.config_7b, "13b": synthetic.config_13b, "tiny": synthetic.config_tiny}[args.model]()
    w = synthetic.device_weights(cfg, dev, seed=0)
    vid, bary = synthetic.body_lift_tables(dev)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=(vid, bary))
    del w
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    S = cfg.sam.img_size

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / args.reps

    for B in (1, 4, 8, 16):
        ic, im = synthetic.images(cfg, dev, seed=5, batch=B)
        t_gen = timeit(lambda: m.generate_batch(ic, [ids[0]] * B, forced_new_tokens=forced))
        t_clip = timeit(lambda: m.encode_images(ic))
        t_sam = timeit(lambda: m.model.visual_model.image_encoder(im.reshape((B * im.shape[1],) + tuple(im.shape[2:]))))
        t_all = timeit(lambda: m.evaluate_batch(ic, im, [ids[0]] * B, [cams[0]] * B, [(S, S)] * B, [(S, S)] * B,
                                                forced_new_tokens=forced))
        m.overlap_sam_encoder = False
        t_ser = timeit(lambda: m.evaluate_batch(ic, im, [ids[0]] * B, [cams[0]] * B, [(S, S)] * B, [(S, S)] * B,
                                                forced_new_tokens=forced))
        m.overlap_sam_encoder = True
        print(f"B={B}: generate_batch {t_gen:.1f} ms (CLIP {t_clip:.1f}), SAM encoder {t_sam:.1f} ms, evaluate_batch "
              f"{t_all:.1f} ms overlapped / {t_ser:.1f} ms serial -> {1e3 * B / t_all:.2f} images/s", flush=True)


if __name__ == "__main__":
    main()
