#!/usr/bin/env python3
"""LLaMA-7B prefill (330 positions): eager launches vs one captured HIP graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import llava, synthetic
    from interactvlm_amd import weights as Wt

    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b().llama
    w = {}
    for k, shape in Wt.llama_spec(cfg).items():
        t = torch.randn(shape, device=dev, dtype=torch.float32)
        w[k] = ((1.0 + 0.05 * t) if (len(shape) == 1) else t / float(shape[-1]) ** 0.5).to(torch.bfloat16)
        del t
    llm = llava.Llama(w, cfg, dev, max_len=1024)
    del w
    T0 = 330
    x = torch.randn(T0, cfg.hidden, device=dev) * 0.5

    def timeit(f, n=10):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            f()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    t_eager = timeit(lambda: llm.forward(x, 0))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        llm.forward(x, 0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = llm.forward(x, 0)
    t_graph = timeit(lambda: g.replay())
    print(f"prefill {T0} positions: eager {t_eager:.2f} ms, graph replay {t_graph:.2f} ms")


if __name__ == "__main__":
    main()
