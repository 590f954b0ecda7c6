# dp64 job on one GPU with 8 and 16 images per evaluate_batch call
for pc in 8 16; do
  python bench.py --workload dp64 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-variants --dp-per-call $pc 2>/dev/null > /tmp/dp_$pc.json
  python - $pc <<'PY'
import json, sys
d = json.loads(open(f"/tmp/dp_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("per call", sys.argv[1], d["value"], "images/s", d["ms_per_step"], "ms per 64")
PY
done
