#!/bin/bash
# [r6] batched decode step (16 / 8 sequences), loop forms of skinny_p12m_kernel: P12M_T=100 product, 101 refill-after-use (2 step pairs in
# flight), 102 (3 - 4 in flight).  Alternating, two rounds.
R=$(cd "$(dirname "$0")/../.." && pwd)
for i in 1 2; do for v in 100 101 102; do for b in 16 8; do
  echo -n "P12M_T=$v "; P12M_T=$v python $R/tools/bench_decode.py --batch $b 2>/dev/null | tail -1
done; done; done
