"""GPU-side rehearsal of the multi-rank step (one GPU: a 1-rank RCCL group issuing every collective of the N-rank step).
The N > 1 exchange itself can only run on the driver's multi-GPU node; CPU (gloo, world size 2) tests are in
test_parallel_cpu.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_captured_step_with_rccl_allreduce(dev):
    """forward + backward + the two RCCL all-reduces (the first started from inside backward) + Adam captured into ONE
    HIP graph; a replay equals an eager step from the same state (tools/debug/rccl_graph_step.py, a fresh child process:
    a process group and a captured collective do not belong in the test runner's process)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = "29547"
    env.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")   # a watchdog complaint must not abort the rehearsal process
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "rccl_graph_step.py"), "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and "OK captured step" in r.stdout, tail
    print(tail.splitlines()[-1])
