"""TargetInterface / estimator getters of the plugin surface and the rt_logger-equivalent snapshots, against the ORACLE's
getters (not the library's own):
  getMeasuredPose target_interface.hpp:130 / .cpp:117-121,142-146;  getPeriodEstimate :94 / .cpp:80-87;
  getEstimatedTransform :106 / .cpp:95-98;  getN / getM :142,148;  getEstimator()->getQ / getR / getP0 kalman.hpp:74-89;
  logger channels measurement / pose / twist / acceleration / covariance target_interface.cpp:32-40, files of
  test/target_manager_test.cpp:164-168 in writeTxtFile's format (utils.hpp:96-120)."""
import numpy as np
import pytest

import oracle
from conftest import HARNESS_ORDER, model_path, synth_stream

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def _spd(A, rng, s=0.3):
    B = rng.normal(size=A.shape) * s
    d = np.sqrt(np.diag(A))
    return A + (B @ B.T) * np.outer(d, d)


@pytest.mark.parametrize("name", HARNESS_ORDER)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_target_interface_getters_match_the_oracle(models, name, dtype):
    m = models[name]
    N, steps, dt = 70, 9, 0.004
    p0, meas = synth_stream(name, N, steps, seed=12, rpy_noise=0.02)
    rng = np.random.default_rng(3)
    ids = rng.permutation(900)[:N].astype(np.uint32)
    mgr = te.TargetManager(model_path(name), dtype=dtype)
    mgr.set_keep_measurement(True)
    mgr.init_batch(ids, dt, 0.0, p0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype)
    # before any measurement: initPose (target_interface.cpp:25)
    ok, mp = mgr.getMeasuredPose(int(ids[3]))
    assert ok
    np.testing.assert_array_equal(mp, [0, 0, 0, 0, 0, 0, 1.0])
    np.testing.assert_array_equal(orc.measured_pose()[3], mp)
    b = mgr.batches()[0]
    for s in range(steps):
        mask = (rng.random(N) < 0.7).astype(np.uint8)
        if s % 3 == 0:      # by-id host path, ids in a shuffled order
            order = rng.permutation(N)
            mgr.update_batch(ids[order], dt, meas[s][order], mask[order])
        elif s % 3 == 1:    # dense device path
            soa = torch.from_numpy(np.ascontiguousarray(meas[s].T)).cuda().to(b.torch_dtype()).contiguous()
            b.step(dt, soa, torch.from_numpy(mask).cuda())
        else:               # the reference's one-target calls
            for i in range(N):
                mgr.update(int(ids[i]), dt, meas[s][i] if mask[i] else None)
        orc.step(dt, meas[s], mask)
    tol = dict(atol=1e-9) if dtype == "f64" else dict(atol=2e-3, rtol=1e-4)
    mo, po, To = orc.measured_pose(), orc.period_estimate(), orc.transform()
    n, mm = oracle.MODEL_DIMS[m["model"]]
    for i in range(N):
        ok, mp = mgr.getMeasuredPose(int(ids[i]))
        assert ok
        if dtype == "f64":
            np.testing.assert_array_equal(mp, mo[i])          # the measurement itself, bit for bit
        else:
            np.testing.assert_array_equal(mp, mo[i].astype(np.float32).astype(np.float64))
        per = mgr.getPeriodEstimate(int(ids[i]))
        if po[i] < 0:
            assert per == -1.0
        else:   # compared as |omega| = 2 pi / period (a slow rotation has a long period and a large absolute error in it)
            assert 2 * np.pi / per == pytest.approx(2 * np.pi / po[i], abs=1e-9 if dtype == "f64" else 2e-3)
        ok, T = mgr.getEstimatedTransform(int(ids[i]))
        assert ok
        np.testing.assert_allclose(T, To[i], **tol)
        np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12 if dtype == "f64" else 1e-6)
        assert mgr.getN(int(ids[i])) == n and mgr.getM(int(ids[i])) == mm
    if name in ("uniform_velocity", "uniform_acceleration"):
        assert (po == -1.0).all()                             # these models never rotate (twist angular part is zero)
    Q, R, P0 = mgr.getModelMatrices(int(ids[5]))
    np.testing.assert_array_equal(Q, m["Q"]); np.testing.assert_array_equal(R, m["R"]); np.testing.assert_array_equal(P0, m["P"])
    # unknown id: every getter says so
    assert mgr.getMeasuredPose(99999)[0] is False and mgr.getPeriodEstimate(99999) is None
    assert mgr.getEstimatedTransform(99999)[0] is False and mgr.getN(99999) == 0 and mgr.getM(99999) == 0
    assert mgr.getModelMatrices(99999) is None
    # measured poses follow their targets through erase (swap-with-last) and growth
    victim = int(ids[2])
    last_before = mgr.getMeasuredPose(int(ids[-1]))[1]
    assert mgr.erase(victim)
    assert mgr.getMeasuredPose(victim)[0] is False
    np.testing.assert_array_equal(mgr.getMeasuredPose(int(ids[-1]))[1], last_before)
    more = np.arange(300, dtype=np.uint32) + 2000
    mgr.init_batch(more, dt, 0.0, np.tile(p0[:1], (300, 1)))
    np.testing.assert_array_equal(mgr.getMeasuredPose(int(ids[-1]))[1], last_before)
    np.testing.assert_array_equal(mgr.getMeasuredPose(2100)[1], [0, 0, 0, 0, 0, 0, 1.0])
    # switched off: the getter reports "not kept"
    mgr.set_keep_measurement(False)
    assert mgr.getMeasuredPose(int(ids[-1]))[0] is False
    mgr.close()


def test_model_matrices_per_class_and_per_target(models):
    """getQ / getR / getP0 give back what each target was created with: classes of one batch, per-target P0, and a second
    model in the same manager."""
    m = models["uniform_acceleration"]
    rng = np.random.default_rng(8)
    NC, N = 5, 60
    Q = np.stack([_spd(m["Q"], rng) for _ in range(NC)]); R = np.stack([_spd(m["R"], rng) for _ in range(NC)])
    P0 = np.stack([_spd(m["P"], rng) for _ in range(NC)])
    cls = rng.integers(0, NC, N).astype(np.uint32)
    ids = np.arange(N, dtype=np.uint32) * 7
    p0 = np.tile([1.0, 2, 3, 0, 0, 0, 1], (N, 1))
    mgr = te.TargetManager()
    mgr.init_batch_classes(ids, 0.004, 0.0, p0, m["model"], Q, R, P0, cls)
    mv = models["uniform_velocity"]
    Pper = np.stack([_spd(mv["P"], rng) for _ in range(10)])
    ids2 = np.arange(10, dtype=np.uint32) + 5000
    mgr.init_batch(ids2, 0.004, 0.0, p0[:10], type=mv["model"], Q=mv["Q"], R=mv["R"], P0=Pper)
    assert mgr.erase(int(ids[0])) and mgr.erase(int(ids2[3]))      # records (and their bookkeeping) move
    for i in range(1, N):
        q, r, p = mgr.getModelMatrices(int(ids[i]))
        np.testing.assert_array_equal(q, Q[cls[i]]); np.testing.assert_array_equal(r, R[cls[i]]); np.testing.assert_array_equal(p, P0[cls[i]])
    for i in range(10):
        if i == 3:
            assert mgr.getModelMatrices(int(ids2[i])) is None
            continue
        q, r, p = mgr.getModelMatrices(int(ids2[i]))
        np.testing.assert_array_equal(q, mv["Q"]); np.testing.assert_array_equal(p, Pper[i])
    mgr.close()


def _rows(path, width):
    a = np.loadtxt(path, ndmin=2)
    assert a.shape[1] == width, (path, a.shape)
    return a


@pytest.mark.parametrize("name", HARNESS_ORDER)
def test_log_channels_match_the_oracle_getters(tmp_path, models, name):
    """Every logged channel against the oracle's own getter after every call: time, measurement, pose7, twist, pose6,
    acceleration, full covariance -- in the files the reference's test writes / its plot script loads, in writeTxtFile's
    format (6 significant digits, one space after every value)."""
    m = models[name]
    N, steps, dt = 5, 6, 0.004
    p0, meas = synth_stream(name, N, steps, seed=21)
    ids = np.array([3, 11, 12, 40, 7], dtype=np.uint32)
    mgr = te.TargetManager(model_path(name))
    mgr.init_batch(ids, dt, 0.0, p0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt)
    mgr.log()                                   # no directory: a no-op, as the reference without LOGGER_ON
    assert not list(tmp_path.iterdir())
    mgr.set_log_directory(tmp_path)
    mgr.set_log_targets([11, 40, 7, 555])       # a selection (555 does not exist: skipped)
    n = oracle.MODEL_DIMS[m["model"]][0]
    want = {k: [] for k in ("time", "meas_pose", "est_pose", "est_twist", "pose", "est_acc", "covariance")}
    for s in range(steps):
        mask = np.array([1, 1, 0 if s == 2 else 1, 1, 1], dtype=np.uint8)
        mgr.update_batch(ids, dt, meas[s], mask)
        orc.step(dt, meas[s], mask)
        mgr.log()
        x, P = orc.state()
        want["time"].append(np.full((N, 1), (s + 1) * dt)); want["meas_pose"].append(orc.measured_pose())
        want["est_pose"].append(orc.pose()); want["est_twist"].append(orc.twist()); want["pose"].append(orc.pose6())
        want["est_acc"].append(orc.acceleration()); want["covariance"].append(P.reshape(N, n * n))
    logged = {11: 1, 40: 3, 7: 4}
    for ch, rows in want.items():
        rows = np.stack(rows)                   # [steps, N, width]
        for tid, i in logged.items():
            got = _rows(tmp_path / ("%s_%d" % (ch, tid)), rows.shape[2])
            assert got.shape[0] == steps
            # the file holds %g with 6 significant digits: compare at that resolution
            atol = 1e-9 * np.abs(rows[:, i]).max() if ch in ("covariance", "pose") else 1e-12
            np.testing.assert_allclose(got, rows[:, i], rtol=6e-6, atol=atol)
        assert not (tmp_path / ("%s_3" % ch)).exists() and not (tmp_path / ("%s_555" % ch)).exists()
    text = (tmp_path / "est_pose_11").read_text().splitlines()
    assert all(ln.endswith(" ") and len(ln.split()) == 7 for ln in text)      # `value << " "` per column (utils.hpp:108-110)
    # target 12 missed its measurement on tick 2: the measurement channel keeps the previous one (measured_pose_ semantics)
    mgr.set_log_targets([12])
    mgr.log()
    np.testing.assert_allclose(_rows(tmp_path / "meas_pose_12", 7)[-1], orc.measured_pose()[2], rtol=6e-6)
    # automatic selection: few targets -> all of them
    mgr.set_log_targets([])
    mgr.log()
    assert (tmp_path / "time_3").exists() and _rows(tmp_path / "time_11", 1).shape[0] == steps + 1
    mgr.close()


def test_log_of_a_large_population_is_one_file_per_channel(tmp_path, models):
    """No selection and more than 64 targets: one <channel>_all file per channel, one write per call, ids in front."""
    name = "angular_velocities"
    m = models[name]
    N, dt = 500, 0.004
    p0, meas = synth_stream(name, N, 2, seed=5)
    ids = (np.arange(N, dtype=np.uint32) * 3 + 2)
    mgr = te.TargetManager(model_path(name))
    mgr.init_batch(ids, dt, 0.0, p0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt)
    mgr.set_log_directory(tmp_path)
    for s in range(2):
        mgr.update_batch(ids, dt, meas[s])
        orc.step(dt, meas[s])
        mgr.log()
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == sorted(c + "_all" for c in ("time", "meas_pose", "est_pose", "est_twist", "pose", "est_acc", "covariance"))
    tw = _rows(tmp_path / "est_twist_all", 7)
    assert tw.shape[0] == 2 * N
    np.testing.assert_array_equal(tw[N:, 0], ids)                 # ascending ids, as std::map iteration
    np.testing.assert_allclose(tw[N:, 1:], orc.twist(), rtol=6e-6, atol=1e-12)
    cov = _rows(tmp_path / "covariance_all", 1 + 144)
    Po = orc.state()[1].reshape(N, 144)
    np.testing.assert_allclose(cov[N:, 1:], Po, rtol=6e-6, atol=1e-9 * np.abs(Po).max())
    mgr.close()


def test_measured_pose_through_fused_and_sequence_launches(models):
    """The measured-pose rows behind multi-tick launches: a temporally fused launch (several ticks in one kernel) and a recorded
    sequence, with availability masks -- every target must end with the LAST measurement it actually had (updateMeasurement
    semantics, src/target_interface.cpp:142-146), targets without any keep the initial pose."""
    from target_estimation_amd.streams import make_stream
    name, N, T, dt = "angular_rates", 700, 6, 0.004
    m = models[name]
    st = make_stream(m["model"], N, T, dt, 17, availability=0.5)
    ref = oracle.stream_fill(m["model"], 17, N, T, dt, availability=0.5)
    has = ref["has_meas"].copy()
    has[:, :5] = 0                                  # five targets never measured
    has_dev = torch.from_numpy(has).cuda()
    ids = np.arange(N, dtype=np.uint32)
    for mode in ("fused", "graph"):
        mgr = te.TargetManager(model_path(name))
        mgr.set_keep_measurement(True)
        mgr.init_batch(ids, dt, 0.0, ref["p0"])
        b = mgr.batches()[0]
        if mode == "fused":
            b.step_fused(dt, st["meas"], has_dev)
        else:
            b.step_sequence(dt, st["meas"], has_dev, use_graph=True)
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], ref["p0"], dt)
        for s in range(T):
            orc.step(dt, ref["meas"][s], has[s])
        mo = orc.measured_pose()
        for i in list(range(8)) + [N // 2, N - 1]:
            np.testing.assert_array_equal(mgr.getMeasuredPose(int(i))[1], mo[i])
        np.testing.assert_array_equal(mgr.getMeasuredPose(2)[1], [0, 0, 0, 0, 0, 0, 1.0])
        assert mgr.getNumberMeasurements(2) == 0 and mgr.getNumberMeasurements(N - 1) == int(has[:, N - 1].sum())
        mgr.close()


def test_initial_covariances_beyond_the_host_mirror(models):
    """getP0 is served from a host mirror of the distinct P0 matrices of a model, up to 4096 of them; a manager fed more than that
    stops mirroring and says so (Q and R, per class, are always there)."""
    import ctypes as C
    from target_estimation_amd import capi
    m = models["uniform_velocity"]
    N = 5000
    P0 = np.tile(m["P"], (N, 1, 1)) * (1.0 + 1e-3 * np.arange(N))[:, None, None]      # 5000 distinct matrices
    ids = np.arange(N, dtype=np.uint32)
    p0 = np.tile([0, 0, 0, 0, 0, 0, 1.0], (N, 1))
    mgr = te.TargetManager()
    mgr.init_batch(ids[:100], 0.004, 0.0, p0[:100], type=m["model"], Q=m["Q"], R=m["R"], P0=P0[:100])
    np.testing.assert_array_equal(mgr.getModelMatrices(57)[2], P0[57])               # within the mirror
    mgr.init_batch(ids[100:], 0.004, 0.0, p0[100:], type=m["model"], Q=m["Q"], R=m["R"], P0=P0[100:])
    assert mgr.getModelMatrices(57) is None                                            # P0 no longer kept ...
    Q = np.empty((6, 6)); R = np.empty((3, 3))
    ok = capi.lib().target_manager_get_model_matrices(mgr.handle, 4321, Q.ctypes.data_as(capi.c_double_p), R.ctypes.data_as(capi.c_double_p), None)
    assert ok                                                                          # ... Q and R still are
    np.testing.assert_array_equal(Q, m["Q"]); np.testing.assert_array_equal(R, m["R"])
    x, P = mgr.get_state_batch(ids[4990:4992])
    np.testing.assert_allclose(P[0], P0[4990], rtol=1e-15)                             # the filter itself got every P0
    del C
    mgr.close()


def test_log_of_an_fp32_manager(tmp_path, models):
    name = "uniform_acceleration"
    m = models[name]
    p0, meas = synth_stream(name, 3, 4, seed=2)
    ids = np.array([1, 2, 3], dtype=np.uint32)
    mgr = te.TargetManager(model_path(name), dtype="f32")
    mgr.init_batch(ids, 0.004, 0.0, p0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, 0.004, dtype="f32")
    mgr.set_log_directory(tmp_path)
    for s in range(4):
        mgr.update_batch(ids, 0.004, meas[s])
        orc.step(0.004, meas[s])
        mgr.log()
    np.testing.assert_allclose(_rows(tmp_path / "est_pose_2", 7)[-1], orc.pose()[1], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(_rows(tmp_path / "meas_pose_2", 7)[-1], orc.measured_pose()[1], rtol=6e-6)
    Po = orc.state()[1][1].ravel()
    np.testing.assert_allclose(_rows(tmp_path / "covariance_2", 81)[-1], Po, rtol=2e-3, atol=2e-3 * np.abs(Po).max())
    mgr.close()
