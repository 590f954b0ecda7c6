#!/usr/bin/env python3
"""Timeline summary of the LAST evaluate() step in a rocprofv3 kernel-trace rocpd DB: per-stream busy time, overlap of the
two streams, GPU idle time.  python tools/rocpd_timeline.py x_results.db"""
import sqlite3
import sys


def union_len(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    # steps are delimited by the lift kernel (last kernel of evaluate)
    lifts = [i for i, r in enumerate(rows) if "lift_plan" in r[0]]
    if len(lifts) < 3:
        print("not enough steps")
        return
    for a, b in ((lifts[-3], lifts[-2]), (lifts[-2], lifts[-1])):
        seg = rows[a + 1: b + 1]
        t0, t1 = seg[0][1], max(r[2] for r in seg)
        by = {}
        for n, s, e, q in seg:
            by.setdefault(q, []).append((s, e, n))
        print(f"step: {len(seg)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy(union) {union_len([(s, e) for _, s, e, _ in seg]) / 1e6:.2f} ms, "
              f"sum of kernel durations {sum(e - s for _, s, e, _ in seg) / 1e6:.2f} ms")
        for q, lst in by.items():
            s0, e1 = lst[0][0], max(x[1] for x in lst)
            print(f"  queue {q}: {len(lst)} kernels, window {(s0 - t0) / 1e6:.2f} .. {(e1 - t0) / 1e6:.2f} ms, busy {union_len([(s, e) for s, e, _ in lst]) / 1e6:.2f} ms, "
                  f"sum {sum(e - s for s, e, _ in lst) / 1e6:.2f} ms; first {lst[0][2][:40]} last {lst[-1][2][:40]}")
        for q, lst in by.items():  # the first operations of each queue (start-up ordering of the two streams)
            for s_, e_, n_ in sorted(lst)[:6]:
                print(f"      q{q} {(s_ - t0) / 1e6:8.3f} .. {(e_ - t0) / 1e6:8.3f} ms  {n_[:60]}")
        # idle gaps on each queue (host-induced bubbles on the critical path show up here)
        for q, lst in by.items():
            lst = sorted(lst)
            gaps = [(lst[i + 1][0] - lst[i][1], lst[i][2][:36], lst[i + 1][2][:36], (lst[i][1] - t0) / 1e6) for i in range(len(lst) - 1)]
            big = sorted([g for g in gaps if g[0] > 30000], reverse=True)
            print(f"  queue {q}: idle between its kernels {sum(max(g[0], 0) for g in gaps) / 1e6:.2f} ms total; gaps > 30 us: "
                  f"{len(big)} ({sum(g[0] for g in big) / 1e6:.2f} ms)")
            for g in big[:12]:
                print(f"      {g[0] / 1e3:8.1f} us at {g[3]:7.2f} ms after {g[1]} before {g[2]}")
        # the same kernels while the other queue is busy vs after it went idle (what the two-stream overlap costs per kernel)
        if len(by) == 2:
            qs = sorted(by, key=lambda q_: len(by[q_]))
            side_end = max(e_ for _, e_, _ in by[qs[0]])
            agg = {}
            for s_, e_, n_ in by[qs[1]]:
                k = (n_.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:], e_ <= side_end)
                a_ = agg.setdefault(k, [0, 0])
                a_[0] += 1
                a_[1] += e_ - s_
            names = sorted({k[0] for k in agg}, key=lambda n_: -sum(agg.get((n_, f), [0, 0])[1] for f in (True, False)))
            print(f"  main-queue kernels while the side queue is busy (until {(side_end - t0) / 1e6:.1f} ms) vs after:")
            for n_ in names[:14]:
                a_, b_ = agg.get((n_, True), [0, 0]), agg.get((n_, False), [0, 0])
                print(f"      {n_[-52:]:<52} during: {a_[0]:5d} x {a_[1] / max(a_[0], 1) / 1e3:7.1f} us   after: {b_[0]:5d} x {b_[1] / max(b_[0], 1) / 1e3:7.1f} us")
        # decode-phase probe: time between consecutive llama_decode_attn kernels
        da = [(s, e) for n, s, e, q in seg if "llama_decode_attn" in n]
        if len(da) > 64:
            import statistics
            gaps = [da[i + 1][0] - da[i][0] for i in range(len(da) - 1)]
            print(f"  decode: layer period median {statistics.median(gaps) / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us over {len(gaps)} layers")


if __name__ == "__main__":
    main()
