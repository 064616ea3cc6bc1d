"""The N > 1 path on CPU: world_size-2 gloo.  Targets are independent, so each rank steps its own
contiguous shard with no collective; the optional gather of poses must reassemble the global
ascending-id order.  The CPU oracle stands in for the per-rank compute (tests only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import model_path, synth_stream


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, name, out_path):
    import oracle
    from target_estimation_amd import dist as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = oracle.load_model_yaml(model_path(name))
    dt = 0.004
    p0, meas = synth_stream(name, n_total, 6, seed=21)
    lo, hi = td.shard_bounds(n_total, rank, world)
    shard = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0[lo:hi], dt)
    for s in range(6):
        shard.step(dt, meas[s][lo:hi])
    local = torch.from_numpy(shard.pose())
    full = td.gather_rows(local, n_total, dst=0)
    every = td.all_gather_rows(local, n_total)
    if rank == 0:
        np.save(out_path, full.numpy())
        assert torch.equal(every, full)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 11])
def test_sharded_run_equals_single_process(tmp_path, n_total):
    import oracle
    name = "angular_velocities"
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_total, name, out), nprocs=2, join=True)
    m = oracle.load_model_yaml(model_path(name))
    p0, meas = synth_stream(name, n_total, 6, seed=21)
    whole = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, 0.004)
    for s in range(6):
        whole.step(0.004, meas[s])
    np.testing.assert_array_equal(np.load(out), whole.pose())


def test_shard_bounds_partition():
    from target_estimation_amd import dist as td
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            edges = [td.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b - a >= d - c >= b - a - 1
            for idx in range(0, n, max(1, n // 50)):
                r = td.owner_of(idx, n, world)
                assert edges[r][0] <= idx < edges[r][1]
