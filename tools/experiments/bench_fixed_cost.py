#!/usr/bin/env python3
"""Fixed cost of a dependent chain of small launches inside a HIP graph (what bounds the decode step besides HBM):
GEMV of shrinking N (K = 4096, fp32 x, with / without the fused RMSNorm), chained x -> y -> x, replayed as a graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    K = 4096
    g = torch.ones(K, device=dev).to(torch.bfloat16)
    for rms in (False, True):
        for N in (64, 256, 1024, 4096, 8192, 16384):
            ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(8)]
            wb = (torch.randn(K, N, device=dev) / N ** 0.5).to(torch.bfloat16) if N >= 8 else None
            x = torch.randn(1, K, device=dev)
            n_chain = 64

            def body():
                h = x
                for i in range(n_chain):
                    y = ops.linear(h, ws[i % 8], rms=(g, 1e-5) if rms else None, out_f32=True)  # [1, N]
                    h = x if wb is None else h  # keep the chain through a dependency on y: add y[0] into x cheaply
                    h = ops.add_rows(x, y[:, :8].repeat(1, K // 8).contiguous()) if False else h
                return y
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                body()
            gr.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                gr.replay()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / (10 * n_chain)
            print(f"rms={rms!s:5s} N={N:6d}: {us:7.2f} us per launch ({N * K * 2 / 1e6:6.1f} MB -> {N * K * 2 / us / 1e6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
