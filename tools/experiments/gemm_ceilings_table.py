#!/usr/bin/env python3
"""profiles/r05_gemm_ceilings.txt from the logs of tools/bench_gemm_ceilings.py (product library + the ablation libraries of
tools/experiments/build_abl.sh).     python tools/gemm_ceilings_table.py <dir with *.json> > profiles/r05_gemm_ceilings.txt"""
import json
import os
import sys

PEAK = 2500.0  # dense bf16 / fp16 MFMA peak, TFLOP/s (MI355X_MICROARCH.md)
COLS = [("product", "everything"), ("noepi", "no epilogue"), ("mfma_lds", "MFMA + LDS reads (no DMA)"), ("mfma_dma", "MFMA + DMA (no LDS reads)"),
        ("mfma_only", "MFMA only"), ("dma_lds", "DMA + LDS reads (no MFMA)"), ("dma_only", "DMA + barriers")]


def main():
    d = sys.argv[1]
    data = {}
    for tag, _ in COLS:
        p = os.path.join(d, f"ceil_{tag}.json")
        if os.path.exists(p):
            data[tag] = json.load(open(p))["rows"]
    shapes = list(data["product"].keys())
    print("# Per-shape ceilings of the hot MFMA GEMMs (default precision: fp16 operands), MI355X, microseconds per launch.")
    print("# Columns: the product kernel, then the same kernel with parts COMPILED OUT (tools/experiments/build_abl.sh; IVLM_ABL_* in")
    print("# gemm_common.h) - every ablated build also has no epilogue.  Best of 3 x 20 launches, SAM shapes warm (back to back), LLaMA")
    print("# shapes with 4 rotating weight copies (cold from HBM, as in the pipeline); split-K shapes include their reduction.")
    print("# derived: epilogue = everything - no epilogue; exposed LDS = no epilogue - (MFMA + DMA); exposed DMA = no epilogue - (MFMA + LDS);")
    print("# frac = algorithmic FLOP / time / 2.5 PFLOP/s for the product kernel and for its MFMA-only skeleton (the ceiling of this tiling).")
    hdr = f"{'shape':64s}" + "".join(f"{t[:12]:>13s}" for t, _ in COLS) + f"{'epilogue':>10s}{'exp.LDS':>9s}{'exp.DMA':>9s}{'frac':>7s}{'ceil':>7s}"
    print(hdr)
    for s in shapes:
        row = {tag: data[tag][s]["us"] for tag in data if s in data[tag]}
        fl = data["product"][s]["tflops"] * data["product"][s]["us"]  # = FLOP / 1e6
        line = f"{s:64s}" + "".join(f"{row.get(t, float('nan')):13.1f}" for t, _ in COLS)
        g = lambda k: row.get(k, float("nan"))
        line += f"{g('product') - g('noepi'):10.1f}{g('noepi') - g('mfma_dma'):9.1f}{g('noepi') - g('mfma_lds'):9.1f}"
        line += f"{fl / g('product') / PEAK:7.3f}{fl / g('mfma_only') / PEAK:7.3f}"
        print(line)
    print()
    for tag, desc in COLS:
        print(f"# {tag:10s} = {desc}")


if __name__ == "__main__":
    main()
