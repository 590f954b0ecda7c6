#!/usr/bin/env python3
"""ms/token of the LLaMA decode step alone (graph replay), 7B or 13B shapes: prefill 330 positions, then timed decode steps.
    python tools/bench_decode.py [--model 13b] [--batch B]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=23)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--fuse", action="store_true", help="attention + o_proj in one launch (opt-in)")
    ap.add_argument("--legacy", action="store_true", help="persistent GEMV kernel also for the M = 1 fp32 rows")
    ap.add_argument("--prefill", type=int, default=330, help="positions in the cache before the timed steps (the headline prompt: 256 image tokens + text)")
    ap.add_argument("--precision", default="f16", help="tower precision: f16 (the default mode of the model) | default (bf16)")
    ap.add_argument("--splits", type=int, default=0, help="split-KV ranges per head (0 = library default)")
    ap.add_argument("--wide", type=int, default=-1, help="ivlm_gemv1_bf12m_tuning: 16-wave blocks up to this many row blocks (-1 = default)")
    ap.add_argument("--tiles", type=int, default=0, help="skinny MFMA tiles per block (0 = automatic)")
    a = ap.parse_args()
    from interactvlm_amd import llava, synthetic
    from interactvlm_amd import weights as Wt
    from interactvlm_amd import _lib

    _lib.load().ivlm_skinny_tuning(a.tiles)
    if a.legacy:
        _lib.load().ivlm_gemv_tuning(-1, 0)
    dev = torch.device("cuda:0")
    cfg = (synthetic.config_7b() if a.model == "7b" else synthetic.config_13b()).llama
    spec = Wt.llama_spec(cfg)
    w = {}
    for k, shape in spec.items():
        t = torch.randn(shape, device=dev, dtype=torch.float32)
        w[k] = ((1.0 + 0.05 * t) if (len(shape) == 1) else t / float(shape[-1]) ** 0.5).to(torch.bfloat16)
        del t
    llm = llava.Llama(w, cfg, dev, max_len=1024)
    llm.fuse_attn_oproj = a.fuse
    if a.precision != "default":
        llm.set_precision(a.precision)
    if os.environ.get("P12M_T"):
        _lib.load().ivlm_gemv16_bf12m_tuning(int(os.environ["P12M_T"]))
    if os.environ.get("PARTS_S"):
        assert _lib.load().ivlm_decode_parts_tuning(int(os.environ["PARTS_S"])) == 0
    if a.wide != -1:
        _lib.load().ivlm_gemv1_bf12m_tuning(a.wide)
    if a.splits:
        assert _lib.load().ivlm_llama_decode_attn_splits(a.splits) == 0
        llm.decode_splitkv = True
    del w
    T0 = a.prefill
    x = (torch.randn(T0, cfg.hidden, device=dev) * 0.5)
    nbytes = sum(L[k].numel() * 2 for L in llm.layers for k in ("qkv", "o", "gu", "down")) + llm.lm_head.numel() * 2
    if a.batch == 1:
        llm.forward(x, 0)
        dg = llm.decode_graph()
        def run():
            dg["pos"].fill_(T0)
            dg["pos64"].fill_(T0)
            if dg.get("fused") is not None:
                for k in ("step", "counters", "status"):
                    dg["fused"][k].zero_()
            for s in range(a.steps):
                dg["graph"].replay()
    else:
        B = a.batch
        kc, vc = llm.batch_cache(B)
        for b in range(B):
            llm.forward(x, 0, cache=(kc[:, b], vc[:, b]))
        dg = llm.decode_graph_batch(B)
        def run():
            dg["pos"].fill_(T0)
            for s in range(a.steps):
                dg["graph"].replay()
    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / (a.reps * a.steps)
    print(f"{a.model} batch {a.batch}: {ms:.3f} ms/token  ({nbytes / ms / 1e9:.2f} TB/s of weight bytes, "
          f"{nbytes / 1e9:.2f} GB per token)", flush=True)
    if dg.get("fused") is not None:
        print("fused status", int(dg["fused"]["status"][0]))


if __name__ == "__main__":
    main()
