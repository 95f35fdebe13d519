"""`python bench.py --gpus N` must start its own N ranks when the scaling driver did not: as a child
process, before any GPU call, relaying the one JSON line and the exit code.  `--dry_launch` runs only that launcher: the
ranks form a gloo group on the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_gpus_2_launches_two_ranks_itself():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry_launch"], capture_output=True,
                       text=True, timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["self_launched"] is True


def test_bench_under_a_driver_launch_does_not_spawn_again():
    """the driver's own form: torch.distributed.run sets WORLD_SIZE, bench.py must then be a plain rank"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry_launch"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["self_launched"] is False


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """more ranks than GPUs on this node: refused before anything is started, non-zero exit, no result line"""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_which_modes_a_run_measures():
    """bench.select_modes: the last mode is the line's value.  N > 1 default: replicated DP, the RCCL halo exchange (secondary)
    and the peer-mapped step (primary); --halo rccl / --partition_adjacency make the RCCL form the line; N = 1 runs one mode."""
    sys.path.insert(0, ROOT)
    import bench
    base = ["--cpu_steps", "0"]

    def modes(argv, world, fits=True):
        old = sys.argv
        sys.argv = ["bench.py"] + base + argv
        try:
            args = bench.parse()
        finally:
            sys.argv = old
        return bench.select_modes(args, world, fits)[1]

    assert modes([], 1) == ["single"] and modes(["--force_peer"], 1) == ["peer"] and modes(["--force_partition"], 1) == ["partition"]
    assert modes([], 8) == ["replicated", "partition", "peer"]
    assert modes(["--skip_rccl"], 8) == ["replicated", "peer"]
    assert modes([], 8, fits=False) == ["partition", "peer"]
    assert modes(["--halo", "rccl"], 8) == ["replicated", "partition"]
    assert modes(["--partition_adjacency"], 8) == ["replicated", "partition_adj"]
    assert modes(["--replicate"], 8) == ["replicated"]
    assert modes(["--partition_only"], 2) == ["partition", "peer"]
