#!/usr/bin/env python3
"""Experiment: weight-streaming rate of the decode GEMVs on a SUBSET of the CUs (CU-masked stream), per library build
(IVLM_LIB_PATH): how many CUs does an HBM-bound stream need?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import ops  # noqa: E402
from exp_cumask import masked_stream  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    bf = torch.bfloat16
    shapes = [("qkv", 12288, 4096, "none", True), ("gu", 22016, 4096, "swiglu", True), ("down", 4096, 11008, "none", False)]
    print("lib", os.environ.get("IVLM_LIB_PATH", "default"))
    for d in (256, 192, 128, 96, 64):
        st = masked_stream([i < d for i in range(ncu)], dev) if d < ncu else torch.cuda.Stream(device=dev)
        line = f"{d:3d} CUs:"
        for name, N, K, act, rms in shapes:
            ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(bf) for _ in range(4)]
            x = torch.randn(1, K, device=dev)
            g = torch.ones(K, device=dev).to(bf)
            kw = dict(act=act, rms=(g, 1e-5) if rms else None, out_f32=True)
            with torch.cuda.stream(st):
                def body():
                    for i in range(32):
                        ops.linear(x, ws[i % 4], **kw)
                body()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    body()
                gr.replay()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                for _ in range(5):
                    gr.replay()
                b.record(st)
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / (5 * 32)
            line += f"  {name} {us:6.1f} us {N * K * 2 / us / 1e6:5.2f} TB/s"
        print(line, flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
