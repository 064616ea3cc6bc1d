"""BASELINE.json configs[3] / configs[4] at their per-GPU size (62 500 + 62 500 targets, two motion models in one
manager) through target_manager_step_sequence_all -- the call bench.py times -- against the ORACLE on a 2 000-target
sample per model: state and covariance, and for configs[4] the fused per-tick sphere query's delta.  (Round 1 only
compared this path with the library's own per-batch calls.)"""
import numpy as np
import pytest

import oracle
from test_gpu_parity import check_state

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def _build(models, parts, dtype, ticks, dt, seed, accel_scene=False):
    from target_estimation_amd.streams import make_stream
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    base, out = 0, []
    for k, (name, n) in enumerate(parts):
        m = models[name]
        st = make_stream(te.MODEL_TYPES[name], n, ticks, dt, seed + 17 * k)
        ids = np.arange(n, dtype=np.uint32) + base
        base += n
        p0 = st["p0"].cpu().numpy()
        v0 = a0 = None
        if accel_scene:   # inbound, accelerating targets so that the sphere query has roots to find
            rng = np.random.default_rng(seed + k)
            d = p0[:, :3] / np.linalg.norm(p0[:, :3], axis=1, keepdims=True)
            v0 = np.concatenate([-d * rng.uniform(1, 6, (n, 1)), np.zeros((n, 3))], 1)
            a0 = np.concatenate([rng.normal(0, 1.0, (n, 3)) + [0, 0, -2.0], np.zeros((n, 3))], 1)
        assert mgr.init_batch(ids, dt, 0.0, p0, v0, a0, type=te.MODEL_TYPES[name], Q=m["Q"], R=m["R"], P0=m["P"]) == n
        out.append(dict(name=name, ids=ids, p0=p0, v0=v0, a0=a0, meas64=st["meas"]))
    batches = mgr.batches()
    assert len(batches) == len(parts)
    meas = [o["meas64"].to(b.torch_dtype()).contiguous() for o, b in zip(out, batches)]
    return mgr, batches, out, meas


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("use_graph", [1, 0])
def test_config3_share_matches_oracle(models, dtype, use_graph):
    """62 500 angular-rates + 62 500 angular-velocities targets (one GPU's share of configs[3])."""
    import bench
    parts = bench.MIXED["cfg4"][1]
    ticks, dt = 8, 0.004
    mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, 20240004)
    half = ticks // 2
    for part in (slice(0, half), slice(half, ticks)):
        mgr.step_sequence_all(dt, [m[part] for m in meas], use_graph=use_graph)
    torch.cuda.synchronize()
    for o, b, mm in zip(info, batches, meas):
        m = models[o["name"]]
        n = len(o["ids"])
        sample = np.sort(np.random.default_rng(1).choice(n, 2000, replace=False))
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], o["p0"][sample], dt, dtype=dtype)
        mh = mm[:, :, torch.from_numpy(sample).cuda()].to(torch.float64).cpu().numpy()     # what the kernel saw
        for s in range(ticks):
            orc.step(dt, np.ascontiguousarray(mh[s].T))
        check_state(mgr, o["ids"][sample], orc, dtype, "%s sample of configs[3]" % o["name"])
        np.testing.assert_array_equal(b.slot_ids()[::997], o["ids"][::997])
        assert mgr.getNumberMeasurements(int(o["ids"][-1])) == ticks
    mgr.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_config4_share_with_fused_query_matches_oracle(models, dtype):
    """62 500 angular-rates + 62 500 uniform-acceleration targets with the own-time sphere query of every target
    inside the step kernels (configs[4]): state AND delta of a 2 000-target sample against the oracle."""
    import bench
    parts = bench.MIXED["cfg5"][1]
    ticks, dt = 6, 0.004
    origin, radius = np.zeros(3), 5.0
    mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, 20240005, accel_scene=True)
    deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in batches]
    poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in batches]
    mgr.step_sequence_all(dt, meas, query=(origin, radius, deltas, poses), use_graph=1)
    torch.cuda.synchronize()
    n_hit = 0
    for o, b, mm, dd, pp in zip(info, batches, meas, deltas, poses):
        m = models[o["name"]]
        n = len(o["ids"])
        sample = np.sort(np.random.default_rng(2).choice(n, 2000, replace=False))
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], o["p0"][sample], dt, 0.0, o["v0"][sample], o["a0"][sample], dtype=dtype)
        mh = mm[:, :, torch.from_numpy(sample).cuda()].to(torch.float64).cpu().numpy()
        for s in range(ticks):
            orc.step(dt, np.ascontiguousarray(mh[s].T))
        check_state(mgr, o["ids"][sample], orc, dtype, "%s sample of configs[4]" % o["name"])
        ok_o, pose_o, delta_o = orc.intersection_pose(ticks * dt, origin, radius)
        d = dd.cpu().numpy()[sample]
        hit, hit_o = d > -1, delta_o > -1
        assert (hit != hit_o).mean() <= (0.0 if dtype == "f64" else 0.01)
        both = hit & hit_o
        rtol = 1e-8 if dtype == "f64" else 5e-4
        np.testing.assert_allclose(d[both], delta_o[both], rtol=rtol, atol=rtol)
        np.testing.assert_allclose(pp.cpu().numpy()[sample][both], pose_o[both], atol=1e-7 if dtype == "f64" else 1e-2)
        n_hit += int(both.sum())
    assert n_hit > 200       # the scene has intersections to find
    mgr.close()
