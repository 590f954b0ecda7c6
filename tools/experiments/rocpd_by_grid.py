#!/usr/bin/env python3
"""Per (kernel, grid) durations of a rocprofv3 rocpd database: the shapes behind one kernel name (e.g. the decode GEMV family).
    python tools/rocpd_by_grid.py x_results.db [name substring]"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    c = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    if gx is None:
        print("columns:", cols)
        return
    q = f"select name, {gx}, {wx}, count(*), avg(duration), min(duration) from kernels group by name, {gx} order by sum(duration) desc"
    print(f"{'kernel':<60} {'grid':>9} {'wg':>5} {'calls':>7} {'avg_us':>8} {'min_us':>8}")
    for name, g, w, n, avg, mn in c.execute(q).fetchall():
        if pat in name:
            print(f"{short(name):<60} {g:>9} {w:>5} {n:>7} {avg/1e3:>8.2f} {mn/1e3:>8.2f}")


if __name__ == "__main__":
    main()
