"""Is the GEMM rate clock / power limited?  (1) isolated launches (50 ms idle before each) against back-to-back launches of SAM
mlp2 (16384 x 1280 x 5120); (2) rocm-smi clock / power samples while the GEMM runs back to back for a few seconds."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from interactvlm_amd import ops

    dev = torch.device("cuda:0")
    M, N, K = 16384, 1280, 5120
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    x = torch.randn(M, N, generator=g).to(dev)
    fn = lambda: ops.linear(a, w, residual=x, out=x)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    iso = []
    for _ in range(10):
        time.sleep(0.05)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        iso.append(s.elapsed_time(e) * 1e3)
    print("isolated launches (us):", " ".join(f"{t:.0f}" for t in iso))
    samples = []
    stop = [False]

    def poll():
        while not stop[0]:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                samples.append(" | ".join(l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l) and "GPU[0]" in l))
            except Exception as ex:  # noqa: BLE001
                samples.append(repr(ex))
            time.sleep(0.3)

    th = threading.Thread(target=poll)
    th.start()
    for chunk in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(2000):
            fn()
        e.record(); torch.cuda.synchronize()
        print(f"back to back, chunk {chunk}: {s.elapsed_time(e) / 2000 * 1e3:.1f} us per launch", flush=True)
    stop[0] = True
    th.join()
    for smp in samples[:12]:
        print("  smi:", smp)


if __name__ == "__main__":
    main()
