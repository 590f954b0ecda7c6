#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown/CSV-like text).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [n_steps] > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:90]


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    print(f"# rocprofv3 --kernel-trace summary of {db}")
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, busy {total/1e6:.2f} ms, span {(span[1]-span[0])/1e6:.2f} ms, "
          f"profiled steps (incl. warmup): {steps}")
    print(f"{'kernel':<88} {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{short(name):<88} {n:>7} {tot/1e6:>10.3f} {avg/1e3:>9.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*tot/total:>6.2f}")


if __name__ == "__main__":
    main()
