#!/usr/bin/env python3
"""Does the SAM ViT-H encoder run faster as TWO half-batches on two HIP streams?  (Round 5.)  The ceiling table says the tile GEMMs of a
4-view pass lose 17 - 39 % to epilogues that all 256 CUs run at the same time (one round of tiles, one block per CU): two independent
2-view passes (own graphs, own buffers) drift apart and one's epilogues overlap the other's MFMAs - at the price of half-filled rounds.
    python tools/experiments/exp_encoder_split.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import sam, synthetic
    from interactvlm_amd import weights as Wt

    dev = torch.device("cuda:0")
    cfg = Wt.IvlmCfg(llama=Wt.LlamaCfg(hidden=256, layers=1, heads=2, inter=512, vocab=32003),
                     clip=Wt.ClipCfg(hidden=128, layers=2, heads=2, inter=256), sam=Wt.SamEncCfg())
    w = synthetic.device_weights(cfg, dev, seed=0)
    encs = [sam.SamImageEncoder(w, cfg.sam, dev) for _ in range(4)]
    for e in encs:  # the default precision mode of the model (fp16 operands + exact q)
        e.precision = "parity"
        e.parity_sites = e.SITES_F16Q
    _, im = synthetic.images(cfg, dev)
    views = im[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    main_s = torch.cuda.current_stream(dev)

    def whole():
        return encs[0](views)

    def split(n):
        per = views.shape[0] // n
        outs = []
        for i in range(n):
            streams[i].wait_stream(main_s)
            with torch.cuda.stream(streams[i]):
                outs.append(encs[i](views[i * per: (i + 1) * per]))
        for i in range(n):
            main_s.wait_stream(streams[i])
        return torch.cat(outs)

    ref = whole()
    for n in (2, 4):
        got = split(n)
        print(f"split {n}: max |d| vs the 4-view pass {float((got.float() - ref.float()).abs().max()):.3e}", flush=True)

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    for rnd in range(3):
        print(f"round {rnd}: 4 views in one pass {timed(whole):7.2f} ms | 2 x 2 views on two streams {timed(lambda: split(2)):7.2f} ms | "
              f"4 x 1 view on four streams {timed(lambda: split(4)):7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
