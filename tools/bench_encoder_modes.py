"""SAM ViT-H encoder alone (4 views), default vs parity precision: wall time per call (graph replay)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from interactvlm_amd import sam, synthetic
    from interactvlm_amd import weights as Wt

    dev = torch.device("cuda:0")
    c = Wt.SamEncCfg()
    cfg = synthetic.config_7b()
    spec = Wt.sam_encoder_spec(c)
    w = {k: v for k, v in synthetic.device_weights(cfg, dev).items() if k in spec}
    enc = sam.SamImageEncoder(w, c, dev)
    _, im = synthetic.images(cfg, dev)
    n = int(os.environ.get("N", "5"))
    if os.environ.get("XCD_MAP") is not None:
        from interactvlm_amd import _lib
        _lib.load().ivlm_attention_xcd_map(int(os.environ["XCD_MAP"]))
    if os.environ.get("WIN_V2") is not None:
        from interactvlm_amd import _lib
        _lib.load().ivlm_attention_window_kernel(int(os.environ["WIN_V2"]))
    for mode in (os.environ.get("MODES", "default,parity").split(",")):
      for rik in ((True, False) if os.environ.get("AB_REL") else (True,)):
        # modes: default | parity | f16 | f16q (the fp16-operand site sets of SamImageEncoder)
        enc.precision = "default" if mode == "default" else "parity"
        enc.parity_sites = {"f16": enc.SITES_F16, "f16q": enc.SITES_F16Q, "parity-fast": enc.PARITY_SITES_FAST}.get(mode, enc.PARITY_SITES)
        enc.rel_in_kernel = rik
        enc.rel_in_kernel_global = os.environ.get("GLOB_TAB", "1") == "1"  # REL 5: the global blocks' rel-pos terms in the kernel
        enc.parity_window_arrays = os.environ.get("PWA", "1") == "1"
        if hasattr(enc, "_graphs"):
            enc._graphs.clear()
        enc(im[0])
        enc(im[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            enc(im[0])
        torch.cuda.synchronize()
        print(f"{mode} (rel-pos terms in the attention kernel: {rik}): {(time.perf_counter() - t0) / n * 1e3:.2f} ms per 4 views")


if __name__ == "__main__":
    main()
