"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (stated here, measured margins in DESIGN.md):
  f64 HIP vs f64 oracle : |dx| <= 1e-10 + 1e-10|x| ; |dP| <= 1e-9 * max|P|   (fma + unpivoted
                          Gauss-Jordan vs mul/add + pivoted LU; (1-K) cancellation ~1e3 on step 1)
  f32 HIP vs f32 oracle : |dx| <= 2e-3 + 1e-4|x| ; |dP| <= 2e-3 * max|P|
  target ids / slot order: exact.
"""
import numpy as np
import pytest

import oracle
from conftest import HARNESS_ORDER, MODEL_FILES, model_path, synth_stream

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")

TOL = {"f64": dict(x_atol=1e-10, x_rtol=1e-10, P_rel=1e-9, out_atol=1e-9),
       "f32": dict(x_atol=2e-3, x_rtol=1e-4, P_rel=2e-3, out_atol=5e-3)}

# 101 = thread per target with symmetric-packed P in HBM (1 + TARGET_LAYOUT_SYMMETRIC_PACKED);
# 201 = axis-separable layout (1 + TARGET_LAYOUT_AXIS_SEPARABLE); 0 = automatic (separable here,
# because the shipped Q, R, P0 do not couple axes)
LANES = {"uniform_velocity": {"f64": [0, 1, 3, 101, 103, 201, 301], "f32": [1, 3, 101, 103, 201, 301]},
         "uniform_acceleration": {"f64": [0, 1, 3, 101, 103, 201, 301], "f32": [1, 3, 101, 103, 201, 301]},
         "angular_rates": {"f64": [0, 3, 6, 103, 106, 201, 301], "f32": [2, 3, 6, 102, 103, 106, 201, 301]},
         "angular_velocities": {"f64": [0, 3, 6, 101, 103, 106, 201, 301], "f32": [1, 3, 6, 101, 103, 106, 201, 301]}}
LAYOUT_OF = {0: "axis_separable_packed", 101: "symmetric_packed", 102: "symmetric_packed", 103: "symmetric_packed",
             106: "symmetric_packed", 201: "axis_separable", 301: "axis_separable_packed"}
CASES = [(m, d, g) for m in HARNESS_ORDER for d in ("f64", "f32") for g in LANES[m][d]]


def check_state(mgr, ids, orc, dtype, what=""):
    x, P = mgr.get_state_batch(ids)
    xo, Po = orc.state()
    t = TOL[dtype]
    scale = np.abs(Po).max(axis=(1, 2), keepdims=True)
    ex = np.abs(x - xo) - (t["x_atol"] + t["x_rtol"] * np.abs(xo))
    eP = np.abs(P - Po) / scale
    assert np.isfinite(x).all() and np.isfinite(P).all(), what
    assert ex.max() <= 0, "%s: x error %.3e over tolerance" % (what, ex.max())
    assert eP.max() <= t["P_rel"], "%s: P error %.3e of scale" % (what, eP.max())
    return np.abs(x - xo).max(), eP.max()


def to_soa(meas_np, batch):
    t = torch.from_numpy(np.ascontiguousarray(meas_np.T)).to("cuda")
    return t.to(batch.torch_dtype()).contiguous()


@pytest.mark.parametrize("name,dtype,lanes", CASES)
def test_dense_batch_matches_oracle(models, name, dtype, lanes):
    """N independent targets, dense device-resident path, with a has_meas mask on some ticks and
    predict-only ticks, at a size that is not a multiple of the tile."""
    m = models[name]
    N, steps, dt = 333, 120, 1.0 / m["frequency"]
    p0, meas = synth_stream(name, N, steps, seed=11)
    rng = np.random.default_rng(5)
    v0 = rng.uniform(-0.5, 0.5, (N, 6)) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
    a0 = rng.uniform(-0.5, 0.5, (N, 6)) * 0.1
    ids = (np.arange(N, dtype=np.uint32) * 7 + 3)
    mgr = te.TargetManager(model_path(name), dtype=dtype, lanes_per_target=lanes)
    assert mgr.init_batch(ids, dt, 0.0, p0, v0, a0) == N
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, 0.0, v0, a0, dtype=dtype)
    b = mgr.batches()[0]
    assert b.size == N and b.lanes_per_target == (lanes % 100 or 1) and b.state_dim == m["Q"].shape[0]
    assert b.layout == LAYOUT_OF.get(lanes, "full")
    np.testing.assert_array_equal(b.slot_ids(), ids)
    check_state(mgr, ids, orc, dtype, "after init")
    worst = (0.0, 0.0)
    for s in range(steps):
        if s % 10 == 7:                       # predict-only tick (TargetInterface::update)
            b.step(dt, None)
            orc.step(dt, None)
        elif s % 10 == 3:                     # masked tick
            mask = rng.random(N) < 0.7
            b.step(dt, to_soa(meas[s], b), torch.from_numpy(mask.astype(np.uint8)).cuda())
            orc.step(dt, meas[s], mask.astype(np.uint8))
        else:
            b.step(dt, to_soa(meas[s], b))
            orc.step(dt, meas[s])
        if s in (0, 1, 9, 49, steps - 1):
            e = check_state(mgr, ids, orc, dtype, "step %d" % s)
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
    # derived outputs (updateTargetState + getters) and extrapolation
    pose, twist, acc, found = mgr.get_est_batch(ids)
    assert found.all()
    t = TOL[dtype]
    np.testing.assert_allclose(pose, orc.pose(), atol=t["out_atol"])
    np.testing.assert_allclose(twist, orc.twist(), atol=t["out_atol"], rtol=1e-6)
    np.testing.assert_allclose(acc, orc.acceleration(), atol=t["out_atol"], rtol=1e-6)
    t1 = steps * dt + 0.1
    pose, twist, acc, _ = mgr.get_est_batch(ids, t1=t1)
    np.testing.assert_allclose(pose, orc.pose_at(t1), atol=t["out_atol"])
    np.testing.assert_allclose(twist, orc.twist_at(t1), atol=t["out_atol"], rtol=1e-6)
    np.testing.assert_allclose(acc, orc.acceleration_at(t1), atol=t["out_atol"], rtol=1e-6)
    dp, dtw, dac = b.get_est()
    np.testing.assert_allclose(dp.cpu().numpy(), orc.pose(), atol=t["out_atol"])
    print("\n[parity] %s %s G=%d worst |dx| %.3e  worst dP/scale %.3e" % (name, dtype, lanes, worst[0], worst[1]))
    mgr.close()


def coupled(m, seed=3):
    """Q, R, P0 that couple every axis with every other one (valid covariances, same scales)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in ("Q", "R", "P"):
        A = m[k]
        n = A.shape[0]
        B = rng.normal(size=(n, n)) * 0.2
        d = np.sqrt(np.diag(A))
        out[k] = A + (B @ B.T) * np.outer(d, d)
    return out


@pytest.mark.parametrize("name,dtype", [(m, d) for m in HARNESS_ORDER for d in ("f64", "f32")])
def test_general_matrices_use_the_dense_kernel(models, name, dtype):
    """With Q, R, P0 that couple the axes the automatic layout must fall back to the dense kernel (and the
    separable layout must be refused): on the upper triangle when the matrices are symmetric, on the full P
    when they are not; parity as for the shipped models."""
    m = models[name]
    cm = coupled(m)
    N, steps, dt = 150, 40, 0.004
    p0, meas = synth_stream(name, N, steps, seed=13)
    ids = np.arange(N, dtype=np.uint32)
    P_asym = cm["P"].copy()
    P_asym[0, 1] *= 0.5                                       # a valid input for the reference: any matrix goes
    for P0, want in ((cm["P"], "symmetric_packed"), (P_asym, "full")):
        mgr = te.TargetManager(dtype=dtype)                   # no default model: typed init
        mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=cm["Q"], R=cm["R"], P0=P0)
        b = mgr.batches()[0]
        assert b.layout == want
        orc = oracle.OracleBatch(m["model"], cm["Q"], cm["R"], P0, p0, dt, dtype=dtype)
        for s in range(steps):
            b.step(dt, to_soa(meas[s], b))
            orc.step(dt, meas[s])
        check_state(mgr, ids, orc, dtype, "coupled %s %s" % (name, want))
        _, P = mgr.get_state_batch(ids[:3])
        n = P.shape[1]
        assert np.abs(P[0][0, 1]) > 0 and np.abs(P[0][1, n - 1]) > 0     # really dense
        mgr.close()
    sep = te.TargetManager(dtype=dtype, lanes_per_target=201)
    with pytest.raises(RuntimeError, match="couple different axes"):
        sep.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=cm["Q"], R=cm["R"], P0=cm["P"])
    sep.close()


@pytest.mark.parametrize("name", ["uniform_acceleration", "angular_velocities"])
def test_automatic_layout_follows_the_matrices(models, name):
    """Shipped (symmetric, axis-separable) matrices -> packed group blocks; a P0 that is separable but not
    symmetric -> full group blocks (nothing may be dropped); parity with the oracle in both cases."""
    m = models[name]
    N, steps, dt = 100, 30, 0.004
    p0, meas = synth_stream(name, N, steps, seed=19)
    ids = np.arange(N, dtype=np.uint32)
    P_asym = m["P"].copy()
    j = 3 if name == "uniform_acceleration" else 6                    # x and its rate: same axis group
    P_asym[0, j] = 0.25 * np.sqrt(P_asym[0, 0] * P_asym[j, j])       # ... coupled on one side only
    for P0, want in ((m["P"], "axis_separable_packed"), (P_asym, "axis_separable")):
        mgr = te.TargetManager(dtype="f64")
        mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=P0)
        b = mgr.batches()[0]
        assert b.layout == want
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], P0, p0, dt, dtype="f64")
        check_state(mgr, ids, orc, "f64", "%s init" % want)
        for s in range(steps):
            b.step(dt, to_soa(meas[s], b))
            orc.step(dt, meas[s])
        check_state(mgr, ids, orc, "f64", want)
        mgr.close()


@pytest.mark.parametrize("name", HARNESS_ORDER)
def test_separable_layout_equals_dense_bit_for_bit(models, name):
    """The axis-separable kernel is the dense arithmetic minus the exact-zero terms: same bits in x
    and in every structurally non-zero entry of P, exact zeros elsewhere (f64)."""
    m = models[name]
    N, steps, dt = 200, 60, 0.004
    p0, meas = synth_stream(name, N, steps, seed=17)
    ids = np.arange(N, dtype=np.uint32)
    res = []
    for lanes in (201, 3 if name in ("uniform_velocity", "uniform_acceleration") else 6):
        mgr = te.TargetManager(model_path(name), lanes_per_target=lanes)
        mgr.init_batch(ids, dt, 0.0, p0)
        b = mgr.batches()[0]
        for s in range(steps):
            b.step(dt, to_soa(meas[s], b), None if s % 7 else torch.from_numpy((np.arange(N) % 3 > 0).astype(np.uint8)).cuda())
        res.append(mgr.get_state_batch(ids))
        mgr.close()
    (xs, Ps), (xd, Pd) = res
    if name == "angular_velocities":          # the EKF Jacobian expressions may contract differently
        np.testing.assert_allclose(xs, xd, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(Ps, Pd, rtol=1e-11, atol=1e-22)
    else:
        np.testing.assert_array_equal(xs, xd)
        np.testing.assert_array_equal(Ps, Pd + 0.0)      # +0.0: -0 == 0


@pytest.mark.parametrize("name", HARNESS_ORDER)
def test_reference_harness_through_the_ten_symbol_abi(models, harness_stream, name):
    """test/target_manager_test.cpp restated against the drop-in C symbols: init, then per step
    update_meas + get_est_pose + get_est_twist, one target, f64, 10 000 steps; the reference's own
    assertions, plus agreement with the oracle and with the committed fixtures."""
    import os
    from conftest import ROOT
    gold = np.load(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"))
    k = HARNESS_ORDER.index(name)
    m = models[name]
    dt = 1.0 / m["frequency"]
    meas = harness_stream[k]
    n_points = meas.shape[0]
    mgr = te.TargetManager(model_path(name), dtype="f64")
    tid = k
    mgr.init(tid, dt, 0.0, meas[0])
    orc = oracle.OracleTarget(m["model"], m["Q"], m["R"], m["P"], meas[0], dt)
    est_pose = np.zeros((n_points, 7))
    est_twist = np.zeros((n_points, 6))
    cps = list(gold["checkpoints"])
    for i in range(n_points):
        mgr.update(tid, dt, meas[i])
        ok, est_pose[i] = mgr.getTargetPose(tid)
        assert ok
        ok, est_twist[i] = mgr.getTargetTwist(tid)
        orc.add_measurement(dt, meas[i])
        if i + 1 in cps:
            j = cps.index(i + 1)
            x, P = mgr.get_state_batch([tid])
            scale = np.abs(gold[name + "_P"][j]).max()
            np.testing.assert_allclose(x[0], gold[name + "_x"][j], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(P[0], gold[name + "_P"][j], rtol=1e-6, atol=1e-8 * scale)
            check_state(mgr, [tid], orc, "f64", "harness step %d" % (i + 1))
            np.testing.assert_allclose(est_pose[i], orc.pose()[0], atol=1e-9)
            np.testing.assert_allclose(est_twist[i], orc.twist()[0], atol=1e-8)
    goal = np.array([0.2, 0.3, 0.4])
    vel = goal / (n_points * dt)
    np.testing.assert_allclose(est_pose[-1, :3], goal, atol=0.01)          # :179-181 etc.
    np.testing.assert_allclose(est_twist[:, :3].mean(0), vel, atol=0.01)   # :187-189 etc.
    if name == "angular_velocities":
        omega = np.array([3.0, 0.01, 0.1])
        np.testing.assert_allclose(est_twist[:, 3:].mean(0), omega, atol=0.05)   # :335-337
        np.testing.assert_allclose(est_twist[-1, 3:], omega, atol=0.01)          # :338-340
    assert mgr.getNumberMeasurements(tid) == n_points
    assert mgr.getTime(tid) == pytest.approx(n_points * dt, rel=1e-12)
    mgr.close()


def test_manager_semantics(models, capfd):
    """Registry behaviour of TargetManager (src/target_manager.cpp:144-295): duplicate init is a
    no-op with a message, unknown ids return false / 0 with a message, ids enumerate ascending,
    erase removes exactly one target, mixed models live in one manager."""
    uv, av = models["uniform_velocity"], models["angular_velocities"]
    mgr = te.TargetManager(model_path("uniform_velocity"))
    p = np.array([1.0, 2.0, 3.0, 0, 0, 0, 1.0])
    for tid in (42, 7, 1000000, 8):
        mgr.init(tid, 0.004, 0.0, p + tid)
    mgr.init(7, 0.004, 0.0, p * 0)            # duplicate: message, no change
    out = capfd.readouterr().out
    assert "Target(7) already exists!" in out
    np.testing.assert_array_equal(mgr.getAvailableTargets(), [7, 8, 42, 1000000])
    ok, pose = mgr.getTargetPose(7)
    assert ok
    np.testing.assert_allclose(pose, [8, 9, 10, 0, 0, 0, 1])
    ok, pose = mgr.getTargetPose(5)
    assert not ok
    mgr.update(5, 0.004, p)
    mgr.update(5, 0.004)
    assert mgr.getNumberMeasurements(5) == 0
    out = capfd.readouterr().out
    assert out.count("Target(5) does not exist!") == 3
    # a second model in the same manager
    q = np.array([0.5, 0.5, 0.5, 0.0, 0.0, np.sin(0.3), np.cos(0.3)])
    mgr.init(9, 0.004, 0.0, q, type=te.ANGULAR_VELOCITIES, Q=av["Q"], R=av["R"], P0=av["P"])
    np.testing.assert_array_equal(mgr.getAvailableTargets(), [7, 8, 9, 42, 1000000])
    assert len(mgr.batches()) == 2
    ok, pose = mgr.getTargetPose(9)
    np.testing.assert_allclose(pose, q, atol=1e-12)
    for s in range(5):
        for tid in (7, 8, 42, 1000000):
            mgr.update(tid, 0.004, p + tid)
        mgr.update(9, 0.004, q)
    assert mgr.getNumberMeasurements(42) == 5 and mgr.getNumberMeasurements(9) == 5
    mgr.update_all(0.004)                      # TargetManager::update(dt): predict everything
    assert mgr.getNumberMeasurements(42) == 5
    assert mgr.getTime(42) == pytest.approx(6 * 0.004) and mgr.getTime(9) == pytest.approx(6 * 0.004)
    # erase: state of the survivors is untouched although slots are compacted
    before = {tid: mgr.get_state_batch([tid]) for tid in (7, 42, 1000000)}
    assert mgr.erase(8) and not mgr.erase(8)
    np.testing.assert_array_equal(mgr.getAvailableTargets(), [7, 9, 42, 1000000])
    for tid, (x, P) in before.items():
        x2, P2 = mgr.get_state_batch([tid])
        np.testing.assert_array_equal(x, x2)
        np.testing.assert_array_equal(P, P2)
    assert mgr.getNumberMeasurements(1000000) == 5
    mgr.init(8, 0.004, 1.5, p)                 # re-create: fresh filter, own clock
    assert mgr.getNumberMeasurements(8) == 0 and mgr.getTime(8) == pytest.approx(1.5)
    mgr.close()


@pytest.mark.parametrize("name", ["uniform_acceleration", "angular_rates"])
def test_by_ids_batch_equals_scalar_calls(models, name):
    """target_manager_update_meas_batch == the loop of target_manager_update_meas, bit for bit."""
    m = models[name]
    N, steps, dt = 50, 12, 0.004
    p0, meas = synth_stream(name, N, steps, seed=3)
    ids = np.arange(100, 100 + N, dtype=np.uint32)
    a = te.TargetManager(model_path(name))
    b = te.TargetManager(model_path(name))
    a.init_batch(ids, dt, 0.0, p0)
    for i, tid in enumerate(ids):
        b.init(tid, dt, 0.0, p0[i])
    perm = np.random.default_rng(0).permutation(N)
    for s in range(steps):
        has = (np.arange(N) + s) % 4 != 0
        a.update_batch(ids[perm], dt, meas[s][perm], has[perm].astype(np.uint8))
        for i in range(N):
            b.update(ids[i], dt, meas[s][i] if has[i] else None)
    xa, Pa = a.get_state_batch(ids)
    xb, Pb = b.get_state_batch(ids)
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(Pa, Pb)
    for tid in ids[:5]:
        assert a.getNumberMeasurements(tid) == b.getNumberMeasurements(tid)
    a.close(); b.close()


def test_duplicate_ids_in_one_batch_call_step_twice(models):
    """An id listed twice in target_manager_update_meas_batch is stepped twice, in order (what the
    reference's loop over ids would do), not raced."""
    name = "uniform_acceleration"
    m = models[name]
    dt = 0.004
    p0, meas = synth_stream(name, 6, 3, seed=4)
    ids = np.arange(6, dtype=np.uint32)
    mgr = te.TargetManager(model_path(name))
    mgr.init_batch(ids, dt, 0.0, p0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt)
    call_ids = np.array([0, 1, 2, 1, 3, 1, 4, 5, 0], dtype=np.uint32)
    call_meas = np.stack([meas[k % 3][i] for k, i in enumerate(call_ids)])
    assert mgr.update_batch(call_ids, dt, call_meas) == len(call_ids)
    for k, i in enumerate(call_ids):                       # the oracle, one target at a time, same order
        one = np.zeros(6, dtype=np.uint8); one[i] = 1
        full = np.zeros((6, 7)); full[i] = call_meas[k]
        # step only target i: emulate with a per-target oracle view
        orc._f("orc_target_add_measurement")(orc._at(int(i)), dt, full[i].ctypes.data_as(oracle.oracle.C.POINTER(oracle.oracle.C.c_double)))
    check_state(mgr, ids, orc, "f64", "duplicates")
    assert mgr.getNumberMeasurements(1) == 3 and mgr.getNumberMeasurements(0) == 2 and mgr.getNumberMeasurements(5) == 1
    mgr.close()
