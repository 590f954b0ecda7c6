#!/bin/bash
# End-to-end A/B of an environment switch on ONE box: `bench.py` b1 (20 steps, no CPU baseline / variants) alternating between the
# two settings for N rounds.  Box-to-box the same build reads 99.8 - 101.4 ms and the first measurement after a pause is slow, so
# only alternating same-box runs can resolve a 1 % effect (round 3: the 256 x 320 tile, the mlp2 K split, the encoder start).
#   tools/ab_e2e.sh VAR A B [rounds] [extra bench.py args...]     e.g.  tools/ab_e2e.sh IVLM_NO_GRAPHS "" 1 3
set -e
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4 2>/dev/null || shift $#
R=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    env "$VAR=$v" python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-variants "$@" 2>/dev/null > /tmp/ab_e2e.json
    python - "$VAR" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_e2e.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]}={sys.argv[2]!r}: {d['value']:.4f} {d['unit']}  {d['ms_per_step']:.3f} ms per step")
PY
  done
done
