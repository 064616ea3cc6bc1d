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


def _gpu_worker(rank, world, port, n_per_model, out_path):
    """One rank of the sharded PRODUCT path: its own HIP TargetManager over its contiguous id range of every model (the
    per-model share SURVEY 8e prescribes), stepped by the call bench.py times, poses gathered to rank 0 in global
    ascending-id order.  Two ranks share the box's one GPU; gloo carries the gather (RCCL needs one GPU per rank)."""
    import target_estimation_amd as te
    from target_estimation_amd import dist as td
    from target_estimation_amd.streams import make_stream
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import oracle
    dt, ticks = 0.004, 5
    mgr = te.TargetManager(dtype="f64")
    meas = []
    base = 0
    for k, name in enumerate(("angular_rates", "angular_velocities")):
        m = oracle.load_model_yaml(model_path(name))   # model file reader only (Q, R, P0)
        lo, hi = td.shard_bounds(n_per_model, rank, world)
        st = make_stream(te.MODEL_TYPES[name], hi - lo, ticks, dt, 500 + k, first_target=lo)   # this rank's slice of the keyed stream
        ids = np.arange(lo, hi, dtype=np.uint32) + base
        base += n_per_model
        mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy(), type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
        meas.append(st["meas"])
    mgr.step_sequence_all(dt, meas, use_graph=0)
    rows = []
    for b in mgr.batches():
        pose, _, _ = b.get_est(twist=False, acc=False)
        rows.append(pose.cpu())
    # per model: gather the shards in id order; models concatenated as the single process enumerates them
    out = []
    for r in rows:
        out.append(td.gather_rows(r, n_per_model, dst=0))
    if rank == 0:
        np.save(out_path, torch.cat(out, 0).numpy())
    dist.barrier()
    dist.destroy_process_group()
    mgr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_per_model", [1001, 4096])
def test_two_hip_shards_equal_the_single_process_hip_run(tmp_path, n_per_model):
    """World size 2 over the PRODUCT: each gloo rank drives its own HIP manager shard on the device; the gathered poses
    equal, bit for bit, the poses of ONE HIP manager holding every target (sharding by contiguous id ranges changes nothing:
    no cross-target term, src/target_manager.cpp:126-133,190-225)."""
    import oracle
    import target_estimation_amd as te
    from target_estimation_amd.streams import make_stream
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_gpu_worker, args=(2, _free_port(), n_per_model, out), nprocs=2, join=True)
    dt, ticks = 0.004, 5
    mgr = te.TargetManager(dtype="f64")
    meas, base = [], 0
    for k, name in enumerate(("angular_rates", "angular_velocities")):
        m = oracle.load_model_yaml(model_path(name))
        st = make_stream(te.MODEL_TYPES[name], n_per_model, ticks, dt, 500 + k)
        ids = np.arange(n_per_model, dtype=np.uint32) + base
        base += n_per_model
        mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy(), type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
        meas.append(st["meas"])
    mgr.step_sequence_all(dt, meas, use_graph=0)
    whole = torch.cat([b.get_est(twist=False, acc=False)[0] for b in mgr.batches()], 0).cpu().numpy()
    got = np.load(out)
    assert got.shape == whole.shape == (2 * n_per_model, 7)
    np.testing.assert_array_equal(got, whole)
    # and the single-process run is the oracle's (a 200-target sample per model)
    for k, name in enumerate(("angular_rates", "angular_velocities")):
        m = oracle.load_model_yaml(model_path(name))
        sample = np.arange(0, n_per_model, max(1, n_per_model // 200))
        ref = oracle.stream_sample(m["model"], 500 + k, sample, ticks, dt)
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], ref["p0"], dt)
        for s in range(ticks):
            orc.step(dt, ref["meas"][s])
        np.testing.assert_allclose(whole[k * n_per_model + sample], orc.pose(), atol=1e-9)
    mgr.close()
