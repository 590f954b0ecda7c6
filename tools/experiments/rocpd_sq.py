#!/usr/bin/env python3
"""Per-kernel SQ counter summary (one rocprofv3 --pmc pass): python tools/rocpd_sq.py x_results.db [name-filter]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = c.execute("select p.name, p.counter_name, avg(p.counter_value), count(*) from pmc_events p "
                     "group by p.name, p.counter_name").fetchall()
    by = {}
    for name, cn, val, n in rows:
        if flt and flt not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))[:70]
        by.setdefault(short, {})[cn] = (val, n)
    for k, d in by.items():
        print(k)
        wc = d.get("SQ_WAVE_CYCLES", (0, 0))[0]
        for cn, (v, n) in sorted(d.items()):
            extra = f"  ({100 * v / wc:.1f} % of WAVE_CYCLES)" if wc and cn.startswith("SQ_") and cn != "SQ_WAVE_CYCLES" else ""
            print(f"   {cn:<28} {v:16.0f}  n={n}{extra}")


if __name__ == "__main__":
    main()
