#!/usr/bin/env python3
"""Per-GEMM-shape time of one evaluate() of the 7B bench workload, kernels run one at a time (two-stream overlap off),
HIP events around every launch.  Prints the optimisation priority list: shape, launches/image, ms/image, TF/s|GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import model as M, ops, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    model = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=synthetic.body_lift_tables(dev))
    del w
    model.overlap_sam_encoder = False
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    step = lambda: model.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)
    for _ in range(2):
        step()
    n = 3
    ops.TIMER.start()
    for _ in range(n):
        step()
    ops.TIMER.stop()
    rows = []
    for fam, unit, scale in (("gemm_bf16_mfma", "TF/s", 1e12), ("gemv_bf16", "GB/s", 1e9)):
        for tag, d in ops.TIMER.by_tag(fam).items():
            rows.append((d["total_s"] / n * 1e3, fam, tag, d["launches"] // n, d["work"] / d["total_s"] / scale, unit))
    rows.sort(reverse=True)
    print(f"{'ms/img':>8} {'family':<16} {'(M,N,K,act)':<34} {'n/img':>6} {'rate':>9}")
    for ms, fam, tag, cnt, rate, unit in rows:
        print(f"{ms:8.3f} {fam:<16} {str(tag):<34} {cnt:6d} {rate:9.1f} {unit}")
    sm = ops.TIMER.summary()
    print(json.dumps({k: {"ms_per_image": round(v["total_s"] / n * 1e3, 3), "launches": v["launches"] // n}
                      for k, v in sm.items()}))


if __name__ == "__main__":
    main()
