#!/usr/bin/env python3
"""HBM roofline of the decode GEMV family from a rocprofv3 --kernel-trace database of the HEADLINE loop (HIP graphs ON: the regime the
throughput figure is measured in - bench.py's own kernel-attached events need eager launches).

    python tools/rocpd_rooflines.py kt_results.db [hidden inter vocab layers]

Bytes per launch are the ALGORITHMIC weight bytes (the bf16 matrix, 2 bytes per weight - SURVEY 8d) of the shape a (kernel, grid) pair
streams; the packed kernels (gemv1_p12m_kernel: 12 bits per weight) MOVE 0.75 of them, which the last column states."""
import re
import sqlite3
import sys

PEAK = 8000.0  # GB/s


def main():
    db = sys.argv[1]
    hidden, inter, vocab, layers = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (4096, 11008, 32003, 32)
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    rows = c.execute(f"select name, {gx}, {wx}, count(*), sum(duration), avg(duration) from kernels where name like '%gemv1_%' "
                     f"group by name, {gx}").fetchall()
    shapes = {3 * hidden: ("q|k|v", hidden), 2 * inter: ("gate|up", hidden), vocab: ("lm_head", hidden)}
    tot_b = tot_t = 0.0
    print(f"# decode GEMV family, graphs ON ({db})")
    print(f"{'kernel':<34} {'matrix':>9} {'launches':>9} {'avg_us':>8} {'MB/launch':>10} {'GB/s':>8} {'of 8 TB/s':>10} {'moved':>6}")
    for name, g, w, n, tot, avg in sorted(rows, key=lambda r: -r[4]):
        m = re.search(r"(gemv1_\w+)<([^>]*)>", name)
        if not m:
            continue
        packed = "p12" in m.group(1)
        nrows = (g // w) * 16  # 16 rows per block in every form (bf16: one per wave of a 1024-thread block; packed: 16-row blocks)
        if nrows in shapes or nrows - 16 < vocab <= nrows:
            label, K = shapes.get(nrows, ("lm_head", hidden))
            nrows = vocab if label == "lm_head" else nrows
            b = nrows * K * 2
        elif nrows == hidden:  # o_proj (K = hidden) and down_proj (K = inter) share a grid: told apart by the packed form, else averaged
            if packed and ", 16, 8" in m.group(2):
                label, b = "down", hidden * inter * 2
            elif packed and inter > 4096:
                label, b = "o", hidden * hidden * 2
            else:
                label, b = "o+down", (hidden * hidden + hidden * inter)
        else:
            continue
        gbs = b / avg  # bytes per ns = GB/s
        tot_b += b * n
        tot_t += tot
        kn = f"{m.group(1)}<{m.group(2)}>"
        print(f"{kn:<34} {label:>9} {n:>9} {avg / 1e3:>8.2f} {b / 1e6:>10.1f} {gbs:>8.0f} {gbs / PEAK:>10.3f} {'0.75' if packed else '1.0':>6}")
    if tot_t:
        print(f"{'family (time-weighted)':<34} {'':>9} {'':>9} {'':>8} {'':>10} {tot_b / tot_t:>8.0f} {tot_b / tot_t / PEAK:>10.3f}")


if __name__ == "__main__":
    main()
