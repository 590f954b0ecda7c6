#!/usr/bin/env python3
"""HBM roofline of the decode GEMV family from a rocprofv3 --kernel-trace database of the HEADLINE loop (HIP graphs ON: the regime the
throughput figure is measured in - bench.py's own kernel-attached events need eager launches).

    python tools/rocpd_rooflines.py kt_results.db [hidden inter vocab layers]

Bytes per launch are the weight bytes of the launches a kernel name covers (7B: gemv1_kernel<true,1> = q|k|v and gate|up,
gemv1_kernel<false,1> = o, down and lm_head)."""
import re
import sqlite3
import sys

PEAK = 8000.0  # GB/s


def main():
    db = sys.argv[1]
    hidden, inter, vocab, layers = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (4096, 11008, 32003, 32)
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels where name like '%gemv1_kernel%' group by name").fetchall()
    qkv, gu = 3 * hidden * hidden * 2, 2 * inter * hidden * 2
    o, down, lm = hidden * hidden * 2, inter * hidden * 2, vocab * hidden * 2
    fam = {"true": (layers * (qkv + gu)) / (2 * layers), "false": (layers * (o + down) + lm) / (2 * layers + 1)}
    tot_b = tot_t = 0.0
    print(f"# decode GEMV family, graphs ON ({db})")
    print(f"{'kernel':<40} {'launches':>9} {'avg_us':>8} {'MB/launch':>10} {'GB/s':>8} {'frac of 8 TB/s':>15}")
    for name, n, tot, avg in rows:
        m = re.search(r"gemv1_kernel<(true|false), (\d)>", name)
        if not m:
            continue
        b = fam[m.group(1)]
        gbs = b / avg  # bytes per ns = GB/s
        tot_b += b * n
        tot_t += tot
        print(f"gemv1_kernel<{m.group(1)}, {m.group(2)}>{'':<20} {n:>9} {avg / 1e3:>8.2f} {b / 1e6:>10.1f} {gbs:>8.0f} {gbs / PEAK:>15.3f}")
    if tot_t:
        print(f"{'family (time-weighted)':<40} {'':>9} {'':>8} {'':>10} {tot_b / tot_t:>8.0f} {tot_b / tot_t / PEAK:>15.3f}")


if __name__ == "__main__":
    main()
