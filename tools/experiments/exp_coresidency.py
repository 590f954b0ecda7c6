"""Do a GEMM stream and a GEMV stream run CONCURRENTLY when their blocks fit on a CU together?  SAM mlp1-shaped GEMMs (16384 x 5120 x
1280) on stream A with a forced tile - 512: the 8-phase 256^2 kernel (2 x 230 registers per SIMD + 128 KB of LDS: nothing else fits),
128: 128^2 tiles (2 blocks per CU, 2 x 120 registers per SIMD: a 1024-thread GEMV block of 64 registers fits next to them) - and
the decode GEMVs of one LLaMA-7B layer on stream B: each alone, then both together (time until each stream's own work is done)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    bf = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(torch.bfloat16).to(dev)
    xn, w1, b1 = bf(16384, 1280), bf(5120, 1280), bf(5120)
    h = torch.empty(16384, 5120, dtype=torch.bfloat16, device=dev)
    ws = [bf(12288, 4096), bf(4096, 4096), bf(22016, 4096), bf(4096, 11008)]  # q|k|v, o, gate|up, down: 404 MB per layer
    x4, x11 = torch.randn(1, 4096, generator=g).to(dev), torch.randn(1, 11008, generator=g).to(dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    NG, NV = 60, 150

    def gemms():
        for _ in range(NG):
            ops.linear(xn, w1, b1, act="gelu", out=h)

    def gemvs():
        for _ in range(NV):
            for w in ws:
                ops.linear(x11 if w.shape[1] == 11008 else x4, w, out_f32=True)

    def run(do_a, do_b):
        torch.cuda.synchronize()
        ea, eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)], [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        if do_a:
            with torch.cuda.stream(sa):
                ea[0].record(); gemms(); ea[1].record()
        if do_b:
            with torch.cuda.stream(sb):
                eb[0].record(); gemvs(); eb[1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        return (ea[0].elapsed_time(ea[1]) if do_a else 0.0), (eb[0].elapsed_time(eb[1]) if do_b else 0.0), wall

    for tile in (512, 128, 256):
        lib.ivlm_gemm_tile_override(tile)
        run(True, True)
        a_alone = run(True, False)[0]
        b_alone = run(False, True)[1]
        a_t, b_t, wall = run(True, True)
        print(f"GEMM tile {tile}: GEMMs alone {a_alone:.1f} ms, GEMVs alone {b_alone:.1f} ms (sum {a_alone + b_alone:.1f}); together: GEMM stream "
              f"{a_t:.1f} ms, GEMV stream {b_t:.1f} ms, wall {wall:.1f} ms", flush=True)
    lib.ivlm_gemm_tile_override(0)


if __name__ == "__main__":
    main()
