"""Error and cost of the fp16-operand mode ("f16": every MFMA operand of the three towers as IEEE fp16, one pass) against the
all-parity result (which sits < 1e-5 from the fp32 CPU oracle): several weight / image seeds, the headline 7B shape and
tests/test_parity_mode_gpu.py's full-depth shape with a width-1024 LLaMA.  MODES / SHAPES / SEEDS from the environment."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    modes = os.environ.get("MODES", "bf16,f16,default,parity-fast").split(",")
    for name in os.environ.get("SHAPES", "small-llm,7b").split(","):
        cfg = shapes[name]
        ids, forced = synthetic.prompt_ids(cfg)
        S = cfg.sam.img_size
        for seed in [int(s) for s in os.environ.get("SEEDS", "3,11").split(",")]:
            w = synthetic.device_weights(cfg, dev, seed=seed)
            m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
            del w
            ic, im = synthetic.images(cfg, dev, seed=seed + 2)
            ev = lambda: m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].float().cpu()
            m.set_precision("parity")
            ref = ev()
            line = f"{name} seed {seed}:"
            for mode in modes:
                m.set_precision(mode)
                got = ev()
                ev()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ev()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 5 * 1e3
                d = (got - ref).abs()
                line += f"  {mode} {float(d.max()):.2e} (rms {float(d.pow(2).mean().sqrt()):.1e}) {ms:.1f} ms"
            print(line, flush=True)
            if os.environ.get("COMBOS"):  # which tower owns what is left: one tower at a time in fp16, the others in parity precision
                enc = m.model.visual_model.image_encoder
                line = f"{name} seed {seed} combos:"
                for label, (e_, c_, l_) in (("enc-f16", ("f16", "parity", "parity")), ("clip-f16", ("parity", "f16", "parity")),
                                            ("llm-f16", ("parity", "parity", "f16")), ("enc-f16attn", ("f16attn", "parity", "parity")),
                                            ("enc-f16mlp", ("f16mlp", "parity", "parity")), ("lang-bf16", ("parity", "default", "default"))):
                    m.set_precision("parity")
                    m.vision_tower.precision = c_
                    m.llm.set_precision(l_)
                    enc.parity_sites = {"f16": enc.SITES_F16, "parity": enc.PARITY_SITES,
                                        "f16attn": frozenset(("f16attn", "n2", "h")), "f16mlp": enc.PARITY_SITES_FAST}[e_]
                    d = (ev() - ref).abs()
                    line += f"  {label} {float(d.max()):.2e}"
                print(line, flush=True)
            del m
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
