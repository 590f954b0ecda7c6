"""GPU-side checks of the multi-process path that a single-GPU box can run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graph_capture_next_to_live_rccl_communicator(hip_lib, cuda):
    """Every rank of `bench.py --gpus N` captures its HIP graphs (CLIP tower, decode step) while an RCCL communicator and its
    watchdog thread exist; a world-size-1 group reproduces that, including the all-gather of the contacts."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_graph_with_rccl.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "graphs + RCCL communicator: ok" in r.stdout
