"""The N>1 path of bench.py (batch sharding, gradient all-reduce, max-over-ranks timing, one JSON line from rank 0),
rehearsed with two ranks that share the one GPU of the test box over gloo.  RCCL needs one GPU per rank and is
exercised by the 8-GPU scaling runs only."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, SHIVAE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config",
           "cfg1", "--batch", "16", "--T", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "strong"
    assert j["config"]["global_batch"] == 16 and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and j["n_elbo"] == j["n_elbo"]  # finite
    assert j["roofline"] is not None and "cpu_baseline" not in j
