"""bench.py's rank start-up: `--gpus N` must really run N ranks (round-1 finding: the flag was parsed and ignored).

CPU: the dry run (no device work: spawn, rendezvous, the all-reduce that counts the ranks, the barrier-bracketed
repetitions, the JSON line).  GPU: the real bench, two ranks on the one GPU of the box over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_flag_starts_that_many_ranks():
    p, line = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["value"] is None and "dry-run" in line["data"]       # a dry run never reports a number
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1   # ONE line, from rank 0


def test_single_rank_dry_run():
    p, line = _run(["--steps", "8", "--warmup", "2", "--dry-run"])
    assert p.returncode == 0 and line["n_gpus"] == 1


def test_mismatch_between_flag_and_launcher_fails():
    # the launcher (here: a faked environment) started one rank, the flag says two
    p, line = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None
    assert "WORLD_SIZE=1" in p.stderr


def test_driver_style_launch_is_accepted():
    """The driver's own command for N > 1: torch.distributed.run starts the ranks, bench.py must not spawn again."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"]
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.gpu
def test_two_ranks_end_to_end_on_one_gpu():
    """The literal `bench.py --gpus 2` (it starts its two ranks itself), gloo between them, both on the box's one GPU:
    every rank steps its own shard, rank 0 prints the one line."""
    p, line = _run(["--gpus", "2", "--steps", "8", "--warmup", "2", "--reps", "2", "--workload", "cfg4_64", "--extra-multi", "uv1m_strong",
                    "--extra-steps", "4"], env={"TE_BENCH_BACKEND": "gloo"}, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["targets_total"] == 2 * line["config"]["targets_per_gpu"]
    assert line["extra"][0]["name"] == "uv1m_strong" and line["extra"][0]["targets_per_gpu"] == 500_000
