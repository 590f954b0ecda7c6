"""How much margin does the "parity-encoder" mode keep under 1e-3?  Several weight / image seeds, two model shapes (the headline
7B one and tests/test_parity_mode_gpu.py's full-depth shape with a width-1024 LLaMA): max |dp| of evaluate() against the all-parity
result (which sits < 1e-5 from the fp32 oracle) for the all-split encoder sites and for the fp16-MLP sites."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import model as M
    from interactvlm_amd import synthetic
    from interactvlm_amd import weights as Wt

    dev = torch.device("cuda:0")
    small = Wt.IvlmCfg(llama=Wt.LlamaCfg(hidden=1024, layers=32, heads=8, inter=2752, vocab=32003),
                       clip=Wt.ClipCfg(hidden=256, layers=24, heads=4, inter=512), sam=Wt.SamEncCfg())
    shapes = {"small-llm": small, "7b": synthetic.config_7b()}
    tables = synthetic.body_lift_tables(dev)
    cams = synthetic.human_cam_params()
    for name in os.environ.get("SHAPES", "small-llm,7b").split(","):
        cfg = shapes[name]
        ids, forced = synthetic.prompt_ids(cfg)
        S = cfg.sam.img_size
        for seed in [int(s) for s in os.environ.get("SEEDS", "3,11,12,13").split(",")]:
            w = synthetic.device_weights(cfg, dev, seed=seed)
            m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
            del w
            enc = m.model.visual_model.image_encoder
            ic, im = synthetic.images(cfg, dev, seed=seed + 2)
            ev = lambda: m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].float().cpu()
            m.set_precision("parity")
            ref = ev()
            m.set_precision("default")
            line = f"{name} seed {seed}: default {float((ev() - ref).abs().max()):.2e}"
            m.set_precision("parity-encoder")
            for label, sites in (("all-split", enc.PARITY_SITES), ("f16mlp", enc.PARITY_SITES_FAST)):
                enc.parity_sites = sites
                line += f"  encoder {label} {float((ev() - ref).abs().max()):.2e}"
            m.set_precision("parity-fast")
            line += f"  parity-fast {float((ev() - ref).abs().max()):.2e}"
            for label, lang in (("clip", (1, 0)), ("llm", (0, 1))):  # all-split encoder + one language tower in parity precision
                m.set_precision("parity")
                if not lang[0]:
                    m.vision_tower.precision = "default"
                if not lang[1]:
                    m.llm.set_precision("default")
                line += f"  parity-but-{'llm' if label == 'clip' else 'clip'} {float((ev() - ref).abs().max()):.2e}"
            print(line, flush=True)
            del m, enc
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
