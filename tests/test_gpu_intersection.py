"""Sphere-intersection query (IntersectionSolver, src/intersection_solver.cpp:42-104) on the GPU vs
the oracle.  Parity here is unpinned by the reference (it has no test for the solver, and its root
finder is Eigen's companion-matrix eigen-solver): the oracle's long-double Aberth roots are checked
against numpy.roots in tests/test_oracle_kat.py."""
import numpy as np
import pytest

import oracle
from conftest import model_path

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def scene(N, seed):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = rng.uniform(0.5, 12.0, N)               # some targets start inside the sphere (radius 5)
    p = d * r[:, None]
    v = -d * rng.uniform(1.0, 6.0, N)[:, None] + rng.normal(0, 0.8, (N, 3))   # roughly inbound
    a = rng.normal(0, 1.0, (N, 3)) + np.array([0, 0, -2.0])
    p0 = np.concatenate([p, np.tile([0, 0, 0, 1.0], (N, 1))], 1)
    v0 = np.concatenate([v, rng.normal(0, 0.2, (N, 3))], 1)
    a0 = np.concatenate([a, rng.normal(0, 0.05, (N, 3))], 1)
    return p0, v0, a0


@pytest.mark.parametrize("name,dtype", [("uniform_acceleration", "f64"), ("angular_rates", "f64"),
                                         ("uniform_acceleration", "f32"), ("angular_rates", "f32"),
                                         ("uniform_velocity", "f64"), ("angular_velocities", "f64")])
def test_intersection_matches_oracle(models, name, dtype):
    m = models[name]
    N, dt = 400, 0.004
    p0, v0, a0 = scene(N, 9)
    ids = np.arange(N, dtype=np.uint32)
    mgr = te.TargetManager(model_path(name), dtype=dtype)
    mgr.init_batch(ids, dt, 0.0, p0, v0, a0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, 0.0, v0, a0, dtype=dtype)
    rng = np.random.default_rng(1)
    for s in range(3):                            # a few ticks so the query runs on filtered state
        meas = p0.copy()
        t = (s + 1) * dt
        meas[:, :3] = p0[:, :3] + v0[:, :3] * t + 0.5 * a0[:, :3] * t * t + rng.normal(0, 0.01, (N, 3))
        mgr.update_batch(ids, dt, meas)
        orc.step(dt, meas)
    origin, radius, t1 = np.array([0.1, -0.2, 0.3]), 5.0, 3 * dt + 0.05
    ok_o, pose_o, delta_o = orc.intersection_pose(t1, origin, radius)
    delta, pose, found = mgr.intersect_batch(ids, t1, origin, radius)
    assert found.all()
    if name in ("uniform_velocity", "angular_velocities"):
        # zero acceleration => leading coefficient 0 => "no intersection" (intersection_solver.cpp:6-9)
        assert (delta == -1).all() and (delta_o == -1).all()
        np.testing.assert_array_equal(pose, np.tile([0, 0, 0, 0, 0, 0, 1.0], (N, 1)))
        mgr.close()
        return
    hit_o, hit = delta_o > -1, delta > -1
    assert hit_o.sum() > 50 and (~hit_o).sum() > 50          # the scene exercises both outcomes
    # classification may differ only where a root sits at the |imag| threshold or at delta = 0
    assert (hit != hit_o).mean() <= (0.0 if dtype == "f64" else 0.01)
    both = hit & hit_o
    rtol = 1e-9 if dtype == "f64" else 2e-4
    np.testing.assert_allclose(delta[both], delta_o[both], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(pose[both], pose_o[both], atol=1e-8 if dtype == "f64" else 5e-3)
    # the returned time is a crossing of the sphere
    d = np.linalg.norm(pose[both][:, :3] - origin, axis=1)
    np.testing.assert_allclose(d, radius, atol=1e-7 if dtype == "f64" else 5e-3)
    # scalar entry points agree with the batch, unknown id -> -1 / false
    for i in (0, 1, 2, int(np.argmax(hit))):
        assert mgr.intersection_time(i, t1, origin, radius) == delta[i]
        ok, p7, d1 = mgr.intersection_pose(i, t1, origin, radius)
        assert ok == bool(hit[i]) and d1 == delta[i]
        np.testing.assert_array_equal(p7, pose[i])
    assert mgr.intersection_time(10 ** 6, t1, origin, radius) == -1
    # dense device query at each target's own time (configs[4] per-tick form)
    b = mgr.batches()[0]
    dd, pp = b.intersect_sphere(origin, radius)
    ok2, pose2, delta2 = orc.intersection_pose(3 * dt, origin, radius)
    dd = dd.cpu().numpy()
    same = (dd > -1) == (delta2 > -1)
    assert same.mean() >= (1.0 if dtype == "f64" else 0.99)
    sel = (dd > -1) & (delta2 > -1)
    np.testing.assert_allclose(dd[sel], delta2[sel], rtol=rtol, atol=rtol)
    mgr.close()


@pytest.mark.parametrize("name", ["uniform_acceleration", "angular_rates"])
def test_convergence_gate_matches_oracle(models, name):
    """IntersectionSolver::getIntersectionPoseWithSphere incl. the moving-average gate
    (src/intersection_solver.cpp:91-124), one gate per target, queried every tick while the
    filter runs; window 8 so that the ring wraps several times."""
    m = models[name]
    N, dt, W = 120, 0.004, 8
    p0, v0, a0 = scene(N, 4)
    ids = np.arange(N, dtype=np.uint32)
    mgr = te.TargetManager(model_path(name))
    mgr.init_batch(ids, dt, 0.0, p0, v0, a0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, 0.0, v0, a0)
    gate = oracle.OracleGate(N, W)
    rng = np.random.default_rng(2)
    origin, radius, pos_th, ang_th = np.zeros(3), 5.0, 0.02, 0.02
    n_conv = 0
    for s in range(30):
        t = (s + 1) * dt
        meas = p0.copy()
        meas[:, :3] = p0[:, :3] + v0[:, :3] * t + 0.5 * a0[:, :3] * t * t + rng.normal(0, 0.01, (N, 3))
        mgr.update_batch(ids, dt, meas)
        orc.step(dt, meas)
        ok_o, pose_o, delta_o = orc.intersection_pose(t, origin, radius)
        conv_o, pf_o, af_o = gate.update(ok_o, pose_o, pos_th, ang_th)
        conv, pose, delta, filt = mgr.intersect_converged_batch(ids, t, pos_th, ang_th, origin, radius, filters_length=W)
        np.testing.assert_array_equal(delta > -1, ok_o)
        hit = ok_o
        np.testing.assert_allclose(delta[hit], delta_o[hit], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(filt[hit, 0], pf_o[hit], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(filt[hit, 1], af_o[hit], rtol=1e-6, atol=1e-7)
        # thresholds are compared on values that agree to ~1e-9: allow a flip only right at the threshold
        near = (np.abs(pf_o - pos_th) < 1e-7) | (np.abs(af_o - ang_th) < 1e-6)
        np.testing.assert_array_equal(conv[~near], conv_o[~near])
        assert not conv[~hit].any()
        n_conv += int(conv.sum())
    assert n_conv > 0
    # the reference-order scalar entry (intersection_solver.hpp:98-101) goes through the same gates
    mgr.close()


@pytest.mark.parametrize("dtype,lanes", [("f64", 0), ("f32", 0), ("f64", 3), ("f32", 201), ("f64", 103), ("f32", 3)])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_all_batches_sequence_equals_per_batch_calls(models, dtype, lanes, use_graph):
    """target_manager_step_sequence_all (the batches as concurrent branches of one hipGraph, with the per-tick
    own-time sphere query) == one target_batch_step + one intersect call per batch per tick, bit for bit."""
    from target_estimation_amd.streams import make_stream
    names = ["angular_rates", "uniform_acceleration", "angular_velocities"]
    sizes = [333, 1000, 77]
    ticks, dt = 7, 0.004            # odd: both state mirrors are in use when the sequence ends
    origin, radius = np.array([0.1, -0.2, 0.3]), 5.0

    def build():
        mgr = te.TargetManager(dtype=dtype, lanes_per_target=lanes)   # 3: dense kernel, 3 lanes per target
        mgr.set_stream(torch.cuda.current_stream().cuda_stream)
        base, meas = 0, []
        for k, (name, n) in enumerate(zip(names, sizes)):
            m = models[name]
            p0, v0, a0 = scene(n, 40 + k)
            ids = np.arange(n, dtype=np.uint32) + base
            base += n
            mgr.init_batch(ids, dt, 0.0, p0, v0, a0, type=te.MODEL_TYPES[name], Q=m["Q"], R=m["R"], P0=m["P"])
            st = make_stream(te.MODEL_TYPES[name], n, ticks, dt, 7 + k)
            mm = st["meas"].clone()
            mm[:, :3, :] = torch.as_tensor(p0[:, :3].T.copy(), device="cuda")[None] + 0.01 * mm[:, :3, :]
            meas.append(mm.to(torch.float64 if dtype == "f64" else torch.float32).contiguous())
        return mgr, meas

    ref, meas = build()
    rb = ref.batches()
    assert len(rb) == 3
    for s in range(ticks):
        for j, b in enumerate(rb):
            b.step(dt, meas[j][s])
    want = [b.intersect_sphere(origin, radius) for b in rb]

    mgr, meas2 = build()
    bs = mgr.batches()
    deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in bs]
    poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in bs]
    poses[2] = None                                            # pose output is optional per batch
    half = ticks // 2
    for part in (slice(0, half), slice(half, ticks)):          # two calls: the second one replays nothing stale
        mgr.step_sequence_all(dt, [m[part] for m in meas2], query=(origin, radius, deltas, poses), use_graph=use_graph)
    torch.cuda.synchronize()
    for j in range(3):
        xr, Pr = ref.get_state_batch(np.arange(sizes[j], dtype=np.uint32) + sum(sizes[:j]))
        xg, Pg = mgr.get_state_batch(np.arange(sizes[j], dtype=np.uint32) + sum(sizes[:j]))
        np.testing.assert_array_equal(xg, xr)
        np.testing.assert_array_equal(Pg, Pr)
        np.testing.assert_array_equal(deltas[j].cpu().numpy(), want[j][0].cpu().numpy())
        if poses[j] is not None:
            np.testing.assert_array_equal(poses[j].cpu().numpy(), want[j][1].cpu().numpy())
        assert mgr.getTime(sum(sizes[:j])) == pytest.approx(ticks * dt) and ref.getTime(sum(sizes[:j])) == pytest.approx(ticks * dt)
    assert (deltas[0] > -1).sum() > 10 and (deltas[2] == -1).all()   # AV has no acceleration: never intersects
    assert mgr.getNumberMeasurements(0) == ticks == ref.getNumberMeasurements(0)
    # replay of the recorded graph with the same arguments keeps stepping
    if use_graph:
        mgr.step_sequence_all(dt, [m[half:ticks] for m in meas2], query=(origin, radius, deltas, poses), use_graph=1)
        assert mgr.getTime(0) == pytest.approx((ticks + ticks - half) * dt)
    # wrong number of specs -> error, nothing launched
    with pytest.raises(RuntimeError):
        mgr.step_sequence_all(dt, meas2[:2])
    ref.close(); mgr.close()


def test_gate_moving_averages_meet_the_reference_filter_test(models):
    """test/avg_filter_test.cpp:30-41 on the GPU gate: MovingAvgFilter(1000) fed 10 000 samples of N(5, 1) must end
    with mean within 0.1 of 5 and variance within 0.1 of 1.  The gate's position filter sees |p_k - p_{k-1}|, so the
    intersection poses are laid out along x with N(5,1) increments (all positive at 5 sigma); the angle filter sees
    0.  Also checked: the warm-up rule (mean over the samples seen so far until the window is full) and the
    filtered value against NumPy's window mean, for 64 independent targets at once."""
    N, W, n_meas = 64, 1000, 10000
    name = "uniform_acceleration"
    mgr = te.TargetManager(model_path(name))
    p0 = np.tile([0, 0, 0, 0, 0, 0, 1.0], (N, 1))
    mgr.init_batch(np.arange(N, dtype=np.uint32), 0.004, 0.0, p0)
    b = mgr.batches()[0]
    rng = np.random.default_rng(0)
    inc = rng.normal(5.0, 1.0, (n_meas, N))
    assert inc.min() > 0
    x = np.cumsum(inc, axis=0)
    pose = torch.zeros((N, 7), dtype=torch.float64, device="cuda")
    pose[:, 6] = 1.0
    delta = torch.zeros(N, dtype=torch.float64, device="cuda")      # "an intersection exists" for every target
    xs = torch.from_numpy(x).cuda()
    for k in range(n_meas):
        pose[:, 0] = xs[k]
        want_var = k == n_meas - 1 or k == 499
        conv, filt, var = b.gate_update(delta, pose, 1e9, 1e9, filters_length=W, want_variance=want_var)
        if k in (0, 1, 499, 999, 1000, 5000, n_meas - 1):
            f = filt.cpu().numpy()
            lo = max(0, k + 1 - W)
            np.testing.assert_allclose(f[:, 0], inc[lo:k + 1].mean(axis=0), rtol=1e-10)
            assert (f[:, 1] == 0).all() and conv.all()
        if k == 499:   # the reference's variance while warming up counts the unfilled zeros of the window (utils.hpp:241-246)
            res = inc[:500].mean(axis=0)
            want = (((inc[:500] - res) ** 2).sum(axis=0) + 500 * res ** 2) / 500
            np.testing.assert_allclose(var.cpu().numpy()[:, 0], want, rtol=1e-9)
    mean, variance = filt.cpu().numpy()[:, 0], var.cpu().numpy()[:, 0]
    assert np.abs(mean - 5.0).max() < 0.1 + 0.05        # 64 targets: the worst of 64 draws of a 0.032-sigma mean, the reference tests one
    assert np.abs(mean[0] - 5.0) < 0.1                   # the reference's own assertion, first target (avg_filter_test.cpp:40)
    assert np.abs(variance[0] - 1.0) < 0.1               # avg_filter_test.cpp:41
    assert np.abs(variance - 1.0).max() < 0.2
    mgr.close()


@pytest.mark.parametrize("name", ["uniform_acceleration", "angular_rates"])
def test_intersection_solver_object_has_one_gate_like_the_reference(models, name):
    """The reference's IntersectionSolver is an OBJECT with ONE gate (two moving averages + the previous pose) shared by every
    id queried through it (intersection_solver.hpp:56-126, src/intersection_solver.cpp:19-40,91-124).  C entry points
    target_intersection_solver_*: three targets queried in turn through one solver every tick, against ONE oracle gate fed
    in the same order; a second solver on the same manager has its own, independent gate."""
    import ctypes as C
    from target_estimation_amd import capi
    lib = capi.lib()
    m = models[name]
    N, dt, W = 3, 0.004, 5
    p0, v0, a0 = scene(N, 9)
    ids = np.array([7, 3, 50], dtype=np.uint32)
    mgr = te.TargetManager(model_path(name))
    mgr.init_batch(ids, dt, 0.0, p0, v0, a0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, 0.0, v0, a0)
    gate, gate2 = oracle.OracleGate(1, W), oracle.OracleGate(1, 250)
    s1 = lib.target_intersection_solver_new(mgr.handle, W)
    s2 = lib.target_intersection_solver_new(mgr.handle, 250)       # the reference's default window
    assert s1 and s2
    origin, radius, pos_th, ang_th = np.zeros(3), 5.0, 0.03, 0.03
    op = origin.ctypes.data_as(capi.c_double_p)
    rng = np.random.default_rng(4)
    seen = {True: 0, False: 0}
    pf, af = C.c_double(), C.c_double()
    for s in range(25):
        t = (s + 1) * dt
        meas = p0.copy()
        meas[:, :3] = p0[:, :3] + v0[:, :3] * t + 0.5 * a0[:, :3] * t * t + rng.normal(0, 0.01, (N, 3))
        mgr.update_batch(ids, dt, meas)
        orc.step(dt, meas)
        ok_o, pose_o, delta_o = orc.intersection_pose(t, origin, radius)
        for i in range(N):
            pose = np.full(7, np.nan)
            conv = lib.target_intersection_solver_get_pose_with_sphere(s1, int(ids[i]), t, pos_th, ang_th, op, radius,
                                                                       pose.ctypes.data_as(capi.c_double_p))
            conv_o, pf_o, af_o = gate.update(ok_o[i:i + 1], pose_o[i:i + 1], pos_th, ang_th)
            d = lib.target_intersection_solver_get_time_with_sphere(s1, int(ids[i]), t, op, radius)
            assert (d > -1) == bool(ok_o[i])
            if ok_o[i]:
                assert d == pytest.approx(delta_o[i], rel=1e-8, abs=1e-10)
                np.testing.assert_allclose(pose, pose_o[i], atol=1e-7)
                lib.target_intersection_solver_last_errors(s1, C.byref(pf), C.byref(af))
                assert pf.value == pytest.approx(pf_o[0], rel=1e-6, abs=1e-9) and af.value == pytest.approx(af_o[0], rel=1e-6, abs=1e-7)
                if abs(pf_o[0] - pos_th) > 1e-7 and abs(af_o[0] - ang_th) > 1e-6:
                    assert bool(conv) == bool(conv_o[0])
                    seen[bool(conv)] += 1
            else:
                np.testing.assert_array_equal(pose, [0, 0, 0, 0, 0, 0, 1.0])      # initPose(intersection_pose), :99
                assert not conv
        # the second solver sees target 7 only: its gate is its own
        pose = np.zeros(7)
        conv2 = lib.target_intersection_solver_get_pose_with_sphere(s2, 7, t, pos_th, ang_th, op, radius, pose.ctypes.data_as(capi.c_double_p))
        conv2_o, pf2_o, _ = gate2.update(ok_o[0:1], pose_o[0:1], pos_th, ang_th)
        if ok_o[0] and abs(pf2_o[0] - pos_th) > 1e-7:
            assert bool(conv2) == bool(conv2_o[0])
    assert seen[False] > 0 and seen[True] == 0       # jumping between three targets metres apart keeps the SHARED gate open ...
    # ... and staying on one target closes it once the window holds only that target's (small) errors
    t = 25 * dt
    ok_o, pose_o, _ = orc.intersection_pose(t, origin, radius)
    assert ok_o[1]
    last = None
    for k in range(2 * W):
        pose = np.zeros(7)
        last = lib.target_intersection_solver_get_pose_with_sphere(s1, int(ids[1]), t, pos_th, ang_th, op, radius, pose.ctypes.data_as(capi.c_double_p))
        conv_o, pf_o, af_o = gate.update(ok_o[1:2], pose_o[1:2], pos_th, ang_th)
        assert bool(last) == bool(conv_o[0])
    assert last
    assert lib.target_intersection_solver_get_time_with_sphere(s1, 12345, 0.1, op, radius) == -1.0     # unknown id
    lib.target_intersection_solver_delete(s1)
    lib.target_intersection_solver_delete(s2)
    assert not lib.target_intersection_solver_new(None, 5)
    mgr.close()
