"""N > 1 path on the CPU: the all-to-all exchange layer of grapes_amd.dist over gloo, world_size 2 and 3
(one process per rank, rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_graph_exchange_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for k in range(world):
        assert f"rank {k}/{world} ok" in r.stdout


def test_partition_bounds():
    from grapes_amd.dist import partition_bounds
    for n, w in ((10, 3), (2449029, 8), (7, 8), (111059956, 8)):
        b = partition_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1
