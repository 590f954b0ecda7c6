"""Which tower owns the default mode's full-depth error?  evaluate() on the headline configuration with each tower switched to
"parity" precision separately, compared with the all-parity result (which sits 8e-6 from the fp32 oracle: bench.py), and timed."""
import itertools
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
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=synthetic.body_lift_tables(dev))
    del w
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size

    def run(clip, llm, sam, n=5):
        m.vision_tower.precision = clip
        m.llm.set_precision(llm)
        m.model.visual_model.image_encoder.precision = sam
        ev = lambda: m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"]
        c = ev()
        ev()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ev().cpu()
        torch.cuda.synchronize()
        return c.float().cpu(), (time.perf_counter() - t0) / n * 1e3

    ref, t_ref = run("parity", "parity", "parity")
    print(f"all parity: {t_ref:.1f} ms")
    for clip, llm, sam in itertools.product(("default", "parity"), repeat=3):
        c, t = run(clip, llm, sam)
        print(f"clip={clip:8s} llm={llm:8s} sam={sam:8s}: max|dp| vs all-parity {float((c - ref).abs().max()):.2e}  rms "
              f"{float((c - ref).pow(2).mean().sqrt()):.2e}   {t:.1f} ms/image")


    # ---- inside the SAM encoder: which operand sites carry the error (CLIP / LLaMA in default precision)
    enc = m.model.visual_model.image_encoder
    print("SAM encoder operand sites (clip / llm default):")
    for sites in (("n1", "attn", "proj", "n2", "h"), ("n1", "attn", "proj", "f16mlp"), ("rel32",), ("attn",), ("attn", "proj"), ("n1", "attn", "proj"), ("n2", "h"),
                  ("n1", "n2", "h"), ("attn", "proj", "n2", "h"), ("n1", "attn", "proj", "h"), ("n1", "attn", "proj", "n2")):
        enc.parity_sites = frozenset(sites)
        c, t = run("default", "default", "parity")
        print(f"  sites={','.join(sites):28s}: max|dp| vs all-parity {float((c - ref).abs().max()):.2e}  rms "
              f"{float((c - ref).pow(2).mean().sqrt()):.2e}   {t:.1f} ms/image")


if __name__ == "__main__":
    main()
