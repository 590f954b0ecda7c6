#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases).

    python tools/rocpd_pmc.py fetch.db write.db [out.json]

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KB; on gfx950 FETCH_SIZE
reports HALF the bytes of a wide coalesced stream, so it is doubled (calibrated below on the lm_head GEMV, whose
bytes are known: 32003 x 4096 x 2 B); WRITE_SIZE is taken as is.
"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    q = ("select name, grid_size_x, count(*), avg(counter_value), avg(duration) from ("
         "select p.name as name, p.counter_value as counter_value, p.duration as duration, k.grid_x as grid_size_x "
         "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ?) "
         "group by name, grid_size_x")
    for name, grid, n, val, dur in c.execute(q, (counter,)):
        out[(short(name), grid)] = (n, val, dur)
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for key in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
        n, fk, dur = f[key]
        wk = w.get(key, (0, 0.0, 0))[1]
        rows.append({"kernel": key[0], "grid_x": key[1], "launches": n, "fetch_KB_raw": round(fk, 1),
                     "write_KB": round(wk, 1), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                     "avg_us_under_pmc": round(dur / 1e3, 1)})
    print(f"{'kernel':<60} {'grid':>8} {'n':>6} {'fetchKB(raw)':>13} {'writeKB':>10} {'HBM MB/launch':>14}")
    for r in rows[:40]:
        print(f"{r['kernel'][:60]:<60} {r['grid_x']:>8} {r['launches']:>6} {r['fetch_KB_raw']:>13.1f} {r['write_KB']:>10.1f} "
              f"{r['hbm_bytes_per_launch']/1e6:>14.2f}")
    if len(sys.argv) > 3:
        def agg(prefix):
            sel = [r for r in rows if r["kernel"].startswith(prefix)]
            n = sum(r["launches"] for r in sel)
            return int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel) / max(n, 1)) if sel else None
        json.dump({"gemm_bf16_kernel": agg(("ivlm::gemm_bf16_kernel", "ivlm::gemm256_kernel", "ivlm::gemm320_kernel")),
                   "lift_plan_kernel": agg("lift_plan"),
                   "gemv_kernel": agg(("ivlm::gemv_kernel", "ivlm::gemv1_kernel", "ivlm::gemv1_p12m_kernel", "ivlm::gemv1_p12_kernel")),
                   "note": "avg HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB"},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
