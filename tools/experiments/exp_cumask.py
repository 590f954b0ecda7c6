#!/usr/bin/env python3
"""Experiment: partition the CUs between the HBM-bound decode chain and the MFMA-bound SAM encoder with CU-masked streams
(hipExtStreamCreateWithCUMask) instead of letting the two streams time-slice whole CUs at block granularity."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from interactvlm_amd import ops  # noqa: E402


def masked_stream(bits, dev):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n32 = (len(bits) + 31) // 32
    arr = (ctypes.c_uint32 * n32)()
    for i, b in enumerate(bits):
        if b:
            arr[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(n32), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def main():
    dev = torch.device("cuda:0")
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    print("CUs", ncu)
    bf = torch.bfloat16
    N, K = 12288, 4096
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(bf) for _ in range(6)]
    x = torch.randn(1, K, device=dev)
    A = torch.randn(16384, 1280, device=dev).to(bf)
    Wg = (torch.randn(5120, 1280, device=dev) / 36).to(bf)

    def gemv_loop(n=60):
        for i in range(n):
            ops.linear(x, ws[i % 6], out_f32=True)

    def gemm_loop(n=12):
        for i in range(n):
            ops.linear(A, Wg, act="gelu")

    def timeit(stream, fn):
        with torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            fn()
            b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    full = torch.cuda.Stream(device=dev)
    print(f"GEMV alone, all CUs: {timeit(full, gemv_loop) / 60 * 1e3:.1f} us per launch; GEMM alone: {timeit(full, gemm_loop) / 12 * 1e3:.1f} us")
    pats = {}
    for d in (64, 96, 128):
        pats[f"first{d}"] = [i < d for i in range(ncu)]
        pats[f"mod8<{d * 8 // ncu}"] = [(i % 8) < (d * 8 // ncu) for i in range(ncu)]
        pats[f"mod32<{d * 32 // ncu}"] = [(i % 32) < (d * 32 // ncu) for i in range(ncu)]
    for name, bits in pats.items():
        s1 = masked_stream(bits, dev)
        s2 = masked_stream([not b for b in bits], dev)
        t_v = timeit(s1, gemv_loop) / 60 * 1e3
        t_m = timeit(s2, gemm_loop) / 12 * 1e3
        # both at once
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(s2):
            e[2].record(s2)
            gemm_loop(24)
            e[3].record(s2)
        with torch.cuda.stream(s1):
            e[0].record(s1)
            gemv_loop(120)
            e[1].record(s1)
        torch.cuda.synchronize()
        print(f"{name:10s} ({sum(bits)} CUs decode): GEMV alone {t_v:6.1f} us, GEMM alone on the rest {t_m:7.1f} us | together: GEMV "
              f"{e[0].elapsed_time(e[1]) / 120 * 1e3:6.1f} us, GEMM {e[2].elapsed_time(e[3]) / 24 * 1e3:7.1f} us", flush=True)
    # reference: unmasked two streams
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(sB):
        e[2].record(sB)
        gemm_loop(24)
        e[3].record(sB)
    with torch.cuda.stream(sA):
        e[0].record(sA)
        gemv_loop(120)
        e[1].record(sA)
    torch.cuda.synchronize()
    print(f"unmasked two streams together: GEMV {e[0].elapsed_time(e[1]) / 120 * 1e3:6.1f} us, GEMM {e[2].elapsed_time(e[3]) / 24 * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
