#!/usr/bin/env python3
"""ms/token of the LLaMA-7B decode: persistent one-launch kernel (ivlm_llama_generate) vs the per-op path."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import llava, ops, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    llm = llava.Llama(w, cfg.llama, dev, max_len=640)
    del w
    c = cfg.llama
    T0, n = 330, int(sys.argv[1]) if len(sys.argv) > 1 else 24
    x = (torch.randn(T0, c.hidden, device=dev) * 0.5).to(torch.bfloat16)
    hid = torch.zeros(T0 + n, c.hidden, dtype=torch.bfloat16, device=dev)
    hid[:T0] = llm.forward(x, 0)
    forced = torch.randint(3, 30000, (n,), dtype=torch.int32, device=dev)
    res = {}

    def fused():
        return llm.generate_fused(hid, T0, n, eos=-1, forced=forced)

    def per_op():
        last = hid[T0 - 1: T0]
        for step in range(n):
            ops.argmax(llm.logits(last))
            if step == n - 1:
                break
            last = llm.forward(llm.embed_ids(forced[step: step + 1]), T0 + step)

    def per_op_graph():
        dg = llm.decode_graph()
        dg["pos"].fill_(T0)
        dg["pos64"].fill_(T0)
        if dg.get("fused") is not None:
            dg["fused"]["step"].zero_()
            dg["fused"]["counters"].zero_()
        if dg.get("dataflow") is not None:
            llm.reset_dataflow()
        ops.argmax(llm.logits(hid[T0 - 1: T0]))
        for step in range(n - 1):
            dg["tok"].copy_(forced[step: step + 1])
            dg["graph"].replay()
            hid[T0 + step: T0 + step + 1].copy_(dg["hidden"])

    for name, fn in (("fused", fused), ("per_op", per_op), ("per_op_graph", per_op_graph)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = fn()
        t_enq = (time.perf_counter() - t0) / reps  # host time to enqueue everything (no sync yet)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        per_tok_bytes = 2.0 * (c.layers * (4.0 * c.hidden ** 2 + 3.0 * c.hidden * c.inter) + c.vocab * c.hidden)
        res[name] = {"ms_total": dt * 1e3, "ms_per_token": dt * 1e3 / (n - 1 + 1e-9),
                     "GBps": per_tok_bytes * (n - 1) / dt / 1e9, "host_enqueue_ms_per_token": t_enq * 1e3 / (n - 1 + 1e-9)}
        if name == "fused":
            res[name]["status"] = out[2].cpu().tolist()
    print(json.dumps(res, indent=1))




def trace_report():
    """IVLM_GEN_TRACE=1 python tools/bench_generate.py trace: per-phase time of block 0 (wall_clock64, 100 MHz)."""
    import numpy as np

    ws = ops.llama_generate.last_workspace
    c = synthetic.config_7b().llama
    off = 4096 + 2 * 4096 + sum(((e * 2 + 255) // 256) * 256 for e in (3 * c.hidden, c.hidden, c.hidden, c.hidden, c.hidden, c.inter))
    tr = ws[off: off + (2 * 1000 + 2) * 8].view(torch.int64).cpu().numpy()
    n = int(tr[0])
    ev = tr[1: 1 + 2 * n].reshape(n, 2)
    names = {9: "lm staged", 19: "lm streamed", 29: "lm epilogue", 39: "lm barrier", 50: "attn done", 51: "attn barrier"}
    for ph, nm in enumerate(("qkv", "o", "gu", "down")):
        names.update({10 + ph: nm + " staged", 20 + ph: nm + " streamed", 30 + ph: nm + " epi+prefetch", 40 + ph: nm + " barrier"})
    agg = {}
    for i in range(1, n):
        d = (ev[i, 1] - ev[i - 1, 1]) * 0.01  # us
        agg.setdefault(names.get(int(ev[i, 0]), str(ev[i, 0])), []).append(d)
    tot = 0
    for k, v in agg.items():
        v = np.array(v[2:]) if len(v) > 4 else np.array(v)
        print(f"{k:<20} n={len(v):4d} mean {v.mean():7.2f} us  p50 {np.median(v):7.2f}  max {v.max():7.2f}")
    return agg


if __name__ == "__main__":
    main()
    if os.environ.get("IVLM_GEN_TRACE"):
        trace_report()
