"""In-situ A/B of dispatch choices that were tuned on warm micro-benchmarks: end-to-end evaluate() time (7B shapes, two-stream
overlap on) with one choice flipped at a time.  python tools/bench_variants.py
Caveat: every variant builds a fresh model (fresh graphs) in the same process; the closing "baseline again" line shows how
much the process itself drifts (allocator state, clocks) - differences below that drift are not evidence."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import _lib, ops, synthetic
    from interactvlm_amd import model as M

    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    vid, bary = synthetic.body_lift_tables(dev)
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    lib = _lib.load()

    def run(label, setup=None, teardown=None):
        if setup:
            setup()
        m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=(vid, bary))  # fresh graphs under the flipped choice
        try:
            for _ in range(3):
                m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].cpu()
            torch.cuda.synchronize()
            n = 8
            t = time.perf_counter()
            for _ in range(n):
                m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].cpu()
            torch.cuda.synchronize()
            print(f"{label:<44s} {1e3 * (time.perf_counter() - t) / n:7.2f} ms", flush=True)
        finally:
            del m
            torch.cuda.empty_cache()
            if teardown:
                teardown()

    def setattr_(obj, name, val):
        return lambda: setattr(obj, name, val)

    run("baseline")
    run("split-K off (prefill o/down, CLIP)", setattr_(ops, "SPLITK", False), setattr_(ops, "SPLITK", True))
    run("rel-pos via dot kernel everywhere", setattr_(ops, "RELPOS_GEMM", False), setattr_(ops, "RELPOS_GEMM", True))
    run("GEMM tile 128 everywhere", lambda: lib.ivlm_gemm_tile_override(128), lambda: lib.ivlm_gemm_tile_override(0))
    run("baseline again")


if __name__ == "__main__":
    main()
