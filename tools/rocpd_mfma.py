#!/usr/bin/env python3
"""Per-kernel matrix-core utilisation from rocprofv3 --pmc passes over the headline loop (rocpd databases).

    python tools/rocpd_mfma.py sq.db [grbm.db]

sq.db   : --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 (any subset)
grbm.db : --pmc GRBM_GUI_ACTIVE (optional second pass)
MFMA utilisation of a launch = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x its cycles), the cycles taken (a) from
GRBM_GUI_ACTIVE of the same kernel (second pass) when given and (b) from its duration under the counters at the 2.4 GHz peak clock (a
LOWER bound of the utilisation per cycle: the clock sags to ~1.8 - 2.2 GHz under the MFMA load, MI355X_MICROARCH.md).  Also printed: the
achieved TFLOP/s from SQ_INSTS_VALU_MFMA_MOPS_* when collected (one MOP = 512 FLOP).
"""
import re
import sqlite3
import sys

N_CU, N_SIMD, CLK = 256, 4, 2.4e9


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


def per_kernel(db):
    c = sqlite3.connect(db)
    out = {}
    q = ("select p.name, k.grid_x, p.counter_name, count(*), avg(p.counter_value), avg(p.duration) from pmc_events p "
         "join kernels k on k.dispatch_id = p.dispatch_id group by p.name, k.grid_x, p.counter_name")
    for name, grid, cn, n, val, dur in c.execute(q):
        d = out.setdefault((short(name), grid), {"n": n, "dur": dur})
        d[cn] = val
    return out


def main():
    sq = per_kernel(sys.argv[1])
    gr = per_kernel(sys.argv[2]) if len(sys.argv) > 2 else {}
    rows = []
    for key, d in sq.items():
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not busy or busy < 1e5:
            continue
        dur_s = d["dur"] * 1e-9
        u_time = busy / (N_SIMD * N_CU * CLK * dur_s)
        g = gr.get(key, {}).get("GRBM_GUI_ACTIVE")
        u_grbm = busy / (N_SIMD * N_CU * g) if g else None
        cu = d.get("SQ_BUSY_CU_CYCLES")
        rows.append((d["n"] * d["dur"], key, d, u_time, u_grbm, cu))
    rows.sort(reverse=True)
    tb = sum(d["n"] * d["SQ_VALU_MFMA_BUSY_CYCLES"] for _, _, d, _, _, cu in rows if cu)
    tc = sum(d["n"] * cu for _, _, d, _, _, cu in rows if cu)
    print("# MFMA-pipe utilisation by hardware counter, per kernel of the headline loop (graphs off for counter collection).")
    print("# THE column to read is the last: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = fraction of a busy CU's SIMD-cycles in which")
    print("# the matrix pipe is busy (both counters come from the same SQ instances, so their ratio is scale-free).  The two time-normalised")
    print("# columns are ~32 x low: this rocprofv3 reports the SQ counters of one shader engine of 32, not the chip sum - uncalibrated, kept")
    print("# for the record.  NOT the algorithmic-FLOP fraction of bench.py's roofline_mfma (that one also pays the clock sag, padding rows,")
    print("# the hi + lo second passes and everything that is not an MFMA cycle).")
    if tc:
        print(f"# all kernels below, time-weighted: {tb / (N_SIMD * tc):.3f}")
    print(f"{'kernel':<58} {'grid':>8} {'n':>5} {'us(pmc)':>8} {'MFMA busy / (1024 SIMDs x 2.4 GHz x t)':>40} {'/ GRBM_GUI_ACTIVE':>18} {'/ SQ_BUSY_CU_CYCLES':>20}")
    for _, key, d, ut, ug, cu in rows[:24]:
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        print(f"{key[0][:58]:<58} {key[1]:>8} {d['n']:>5} {d['dur'] / 1e3:>8.1f} {ut:>40.3f} {(f'{ug:.3f}' if ug else '-'):>18} "
              f"{(f'{busy / (N_SIMD * cu):.3f}' if cu else '-'):>20}")


if __name__ == "__main__":
    main()
