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


def _plan(argv, world, fits=True):
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py", "--cpu_steps", "0"] + argv
    try:
        args = bench.parse()
    finally:
        sys.argv = old
    return bench.select_modes(args, world, fits)[1:]


def test_which_modes_a_run_measures():
    """bench.select_modes -> (primary candidates in the order they are tried, secondary measurements run AFTER the primary).
    N > 1 default: the peer-mapped step first (the RCCL form should the mapping be refused), then the replicated DP step and
    the RCCL halo exchange beside it; --halo rccl / --partition_adjacency make the RCCL form the line; N = 1 runs one mode."""
    assert _plan([], 1) == (["single"], []) and _plan(["--force_peer"], 1) == (["peer"], [])
    assert _plan(["--force_partition"], 1) == (["partition"], [])
    assert _plan([], 8) == (["peer", "partition"], ["replicated", "partition"])
    assert _plan(["--skip_rccl"], 8) == (["peer", "partition"], ["replicated"])
    assert _plan([], 8, fits=False) == (["peer", "partition"], ["partition"])
    assert _plan(["--halo", "rccl"], 8) == (["partition"], ["replicated"])
    assert _plan(["--partition_adjacency"], 8) == (["partition_adj"], ["replicated"])
    assert _plan(["--replicate"], 8) == (["replicated"], [])
    assert _plan(["--partition_only"], 2) == (["peer", "partition"], ["partition"])


class _NoGuard:
    def __init__(self, kind, mode):
        self.kind, self.mode = kind, mode

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def test_primary_runs_first_and_a_refused_peer_mapping_makes_the_rccl_form_the_line():
    """the fallback must not be encoded as a mode.  A stubbed run() that raises for 'peer' before any step ran
    (mapping refused on every rank alike): the RCCL form is tried next AS THE PRIMARY, runs exactly once, and is not measured
    again as a secondary; nothing but real mode names ever reaches run()."""
    sys.path.insert(0, ROOT)
    import bench
    prim, sec = _plan([], 8)
    calls = []

    def run(mode):
        calls.append(mode)
        if mode == "peer":
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument")
        return {"mode": mode}
    mode, res, notes = bench.run_primary(prim, run, world=8, peer_unmapped=lambda: True, guard=_NoGuard, log=lambda m: None)
    assert mode == "partition" and res == {"mode": "partition"} and "halo_peer_mapping" in notes
    r2, n2 = bench.run_secondary(sec, run, done={mode: res}, guard=_NoGuard, log=lambda m: None)
    assert calls == ["peer", "partition", "replicated"] and list(r2) == ["replicated"] and not n2


def test_a_primary_failure_after_steps_ran_ends_the_search_and_a_secondary_failure_costs_only_itself():
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    def run(mode):
        calls.append(mode)
        if mode in ("peer", "partition"):
            raise RuntimeError("boom")
        return {"mode": mode}
    # the shards WERE mapped (peer_unmapped() False): the failure came later, other ranks may sit in a collective -> no next candidate
    mode, res, notes = bench.run_primary(["peer", "partition"], run, world=8, peer_unmapped=lambda: False, guard=_NoGuard,
                                         log=lambda m: None)
    assert mode is None and calls == ["peer"] and "peer" in notes
    # a secondary that raises is noted, ends the secondaries, and leaves the primary's result alone
    calls.clear()
    done = {"peer": {"mode": "peer"}}
    r2, n2 = bench.run_secondary(["replicated", "partition", "single"], run, done=done, guard=_NoGuard, log=lambda m: None)
    assert calls == ["replicated", "partition"] and list(r2) == ["replicated"] and "partition" in n2 and done == {"peer": {"mode": "peer"}}
