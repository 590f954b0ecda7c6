"""`python bench.py --gpus N` must launch its own ranks.  The N > 1 form needs N GPUs; the same launcher
path (bench.py -> torch.distributed.run -> one rank per GPU -> RCCL process group -> evaluate_sharded -> one JSON line from
rank 0) is exercised here with a world of ONE rank on the dp64 workload of the tiny configuration."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_self_launch_world1_dp64_tiny(hip_lib, cuda):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--spawn", "--workload", "dp64", "--model", "tiny",
           "--dp-images", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["workload"].startswith("dp64") and d["config"]["images_per_step"] == 16
