"""The ROS node's mailbox / has-measurement / expiry policy (RosTargetManager::update,
src/target_manager_ros.cpp:41-92) as a transport-agnostic ingest stage over the batched GPU path,
against a Python restatement of the policy that drives one CPU-oracle target per id."""
import numpy as np
import pytest

import oracle
from oracle.ingest_policy import RefIngest, get_id
from oracle import np_twin as tw
from conftest import model_path

pytestmark = pytest.mark.gpu
te = pytest.importorskip("target_estimation_amd")


def test_frame_name_parsing():
    m = te.TargetManager(model_path("uniform_velocity"))
    ing = te.MeasurementIngest(m)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for frame, want in [("target_12", 1), ("camera_link", 0), ("base", 0), ("target", -1), ("my_target_3", -1),
                        ("target_x", -1), ("target_7abc", 1), ("xtarget_0", 1)]:
        assert ing.push_named(frame, 1.0, pose) == want, frame
        assert (1 if (("target" in frame) and get_id(frame) is not None) else (0 if "target" not in frame else -1)) == want
    ids, _ = ing.tick(0.004, 1.0)
    np.testing.assert_array_equal(ids, [0, 7, 12])
    ing.close(); m.close()


@pytest.mark.parametrize("name", ["uniform_acceleration", "angular_velocities"])
def test_mailbox_and_expiry_policy_matches_reference_logic(models, name):
    m = models[name]
    dt, n_ticks = 0.004, 160
    rng = np.random.default_rng(3)
    mgr = te.TargetManager()                       # RosTargetManager has no defaults either: typed creation
    ing = te.MeasurementIngest(mgr, m["model"], m["Q"], m["R"], m["P"], expiration_time=0.1)
    ref = RefIngest(m["model"], m["Q"], m["R"], m["P"], expiration_time=0.1)
    ids_all = [3, 11, 12, 40, 41]
    start = {3: 0, 11: 5, 12: 5, 40: 30, 41: 60}          # first tick each target is seen
    stop = {3: 10 ** 9, 11: 80, 12: 10 ** 9, 40: 90, 41: 10 ** 9}   # publisher goes silent -> expiry
    vel = {i: rng.uniform(-1, 1, 3) for i in ids_all}
    om = {i: np.array([rng.uniform(-2, 2), 0.05, -0.05]) for i in ids_all}
    q = {i: np.array([0, 0, 0, 1.0]) for i in ids_all}
    seen = set()
    for k in range(n_ticks):
        now = 100.0 + k * dt
        for i in ids_all:
            if not (start[i] <= k < stop[i]):
                continue
            q[i] = tw.quat_normalize(tw.qtran(dt, om[i]) @ q[i])
            pose = np.concatenate([vel[i] * k * dt + rng.normal(0, 0.01, 3), q[i]])
            # target 12 publishes only every third tick with a fresh stamp, in between it re-sends the
            # old stamp (no new measurement -> predict only); target 3 sometimes sends nothing at all
            if i == 12 and k % 3:
                stamp = 100.0 + (k - k % 3) * dt
            else:
                stamp = now
            if i == 3 and k % 7 == 6:
                continue
            assert ing.push_named("target_%d" % i, stamp, pose) == 1
            ref.push_named("target_%d" % i, stamp, pose)
        ids, poses = ing.tick(dt, now)
        rids, rposes = ref.tick(dt, now)
        np.testing.assert_array_equal(ids, rids)               # creation, expiry, ascending order: exact
        np.testing.assert_allclose(poses, rposes, atol=1e-9)
        seen |= set(ids.tolist())
        for i in rids[:2]:
            assert mgr.getNumberMeasurements(int(i)) == _nmeas(ref.targets[int(i)])
    assert seen == set(ids_all)
    assert set(ids.tolist()) == {3, 12, 41}                     # 11 and 40 timed out and were erased
    x, P = mgr.get_state_batch(ids)
    for j, i in enumerate(ids):
        xo, Po = ref.targets[int(i)].state()
        np.testing.assert_allclose(x[j], xo[0], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(P[j], Po[0], rtol=1e-7, atol=1e-9 * np.abs(Po).max())
    ing.close(); mgr.close()


def _nmeas(orc_target):
    import ctypes as C
    # n_meas is the long long after (model, n, m, initialized, id): offset 24 (see te_oracle.h)
    return C.cast(orc_target.base + 24, C.POINTER(C.c_longlong))[0]


def test_log_snapshots(tmp_path, models):
    """target_manager_log with a log directory: one appended row per target and call, in the text format
    of the reference's writeTxtFile (utils.hpp:96-120)."""
    mgr = te.TargetManager(model_path("uniform_acceleration"))
    ids = np.array([4, 9], dtype=np.uint32)
    p0 = np.array([[1.0, 2, 3, 0, 0, 0, 1], [4, 5, 6, 0, 0, 0, 1]])
    mgr.init_batch(ids, 0.004, 0.0, p0)
    mgr.log()                                   # no directory: no-op
    assert not list(tmp_path.iterdir())
    mgr.set_log_directory(tmp_path)
    for k in range(3):
        mgr.update_batch(ids, 0.004, p0)
        mgr.log()
    est = np.loadtxt(tmp_path / "est_pose_9")
    assert est.shape == (3, 7)
    pose, _, _, _ = mgr.get_est_batch(ids)
    np.testing.assert_allclose(est[-1], pose[1], rtol=1e-5)
    assert np.loadtxt(tmp_path / "covariance_4").shape == (3, 81) and np.loadtxt(tmp_path / "meas_pose_4").shape == (3, 7)
    np.testing.assert_allclose(np.loadtxt(tmp_path / "time_4"), [0.004, 0.008, 0.012], rtol=1e-9)
    mgr.close()
