"""The identities of the reference's test/geometry_test.cpp, applied to the oracle's geometry
subset (tolerance 1e-4 as there), plus C-oracle vs NumPy-twin agreement."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import np_twin as tw


def _call(lib, name, sfx, *arrays, out_shape):
    real = C.c_double if sfx == "f64" else C.c_float
    dt = np.float64 if sfx == "f64" else np.float32
    args = []
    for a in arrays:
        if np.isscalar(a):
            args.append(real(a))
        else:
            a = np.ascontiguousarray(a, dtype=dt)
            args.append(a.ctypes.data_as(C.POINTER(real)))
    out = np.zeros(out_shape, dtype=dt)
    getattr(lib, "%s_%s" % (name, sfx))(*args, out.ctypes.data_as(C.POINTER(real)))
    return out.astype(np.float64)


def random_rotation(rng):
    """generateRandomPose(), test/geometry_test.cpp:4-17: random axis, angle in [-1,1]."""
    axis = rng.uniform(-1, 1, 3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-1, 1)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


@pytest.mark.parametrize("sfx,tol", [("f64", 1e-4), ("f32", 1e-4)])
def test_quat_rpy_round_trips(sfx, tol):
    lib = oracle.load()
    rng = np.random.default_rng(0)
    for _ in range(100):
        R = random_rotation(rng)
        q = _call(lib, "orc_rot_to_quat", sfx, R, out_shape=4)          # q = pose.linear()
        assert abs(np.linalg.norm(q) - 1) < tol
        R2 = _call(lib, "orc_quat_to_rot", sfx, q, out_shape=(3, 3))    # compareToEigenQuat :38-53
        np.testing.assert_allclose(R2, R, atol=tol)
        rpy = _call(lib, "orc_quat_to_rpy", sfx, q, out_shape=3)        # quatToRpyToQuat :55-66
        q2 = _call(lib, "orc_rpy_to_quat", sfx, rpy, out_shape=4)
        np.testing.assert_allclose(q2, q, atol=tol)
        rpy2 = _call(lib, "orc_rot_to_rpy", sfx, R, out_shape=3)        # rot -> rpy == quat -> rpy
        np.testing.assert_allclose(rpy2, rpy, atol=tol)
        # single-axis sanity (singleRotationsRPY :68-104)
    for ang in rng.uniform(-1, 1, 20):
        for ax in range(3):
            rpy = np.zeros(3)
            rpy[ax] = ang
            q = _call(lib, "orc_rpy_to_quat", sfx, rpy, out_shape=4)
            back = _call(lib, "orc_quat_to_rpy", sfx, q, out_shape=3)
            np.testing.assert_allclose(back, rpy, atol=tol)


def test_oracle_geometry_matches_twin():
    lib = oracle.load()
    rng = np.random.default_rng(2)
    for _ in range(200):
        rpy = rng.uniform(-1.4, 1.4, 3) * np.array([2, 1, 2])
        om = rng.uniform(-3, 3, 3)
        dt = rng.uniform(0.001, 0.05)
        np.testing.assert_allclose(_call(lib, "orc_rpy_to_quat", "f64", rpy, out_shape=4), tw.rpy_to_quat(rpy), atol=1e-15)
        q = tw.rpy_to_quat(rpy)
        np.testing.assert_allclose(_call(lib, "orc_quat_to_rpy", "f64", q, out_shape=3), tw.quat_to_rpy(q), atol=1e-14)
        np.testing.assert_allclose(_call(lib, "orc_quat_to_rot", "f64", q, out_shape=(3, 3)), tw.quat_to_rot(q), atol=1e-15)
        np.testing.assert_allclose(_call(lib, "orc_rpy_to_ear_base", "f64", rpy, out_shape=(3, 3)), tw.ear_base(rpy), atol=1e-15)
        np.testing.assert_allclose(_call(lib, "orc_rpy_to_ear_base_inv", "f64", rpy, out_shape=(3, 3)), tw.ear_base_inv(rpy), rtol=1e-14)
        np.testing.assert_allclose(_call(lib, "orc_ear_base_inv_jac_rpy", "f64", rpy, om, dt, out_shape=(3, 3)), tw.jac_rpy(rpy, om, dt), rtol=1e-13, atol=1e-16)
        np.testing.assert_allclose(_call(lib, "orc_ear_base_inv_jac_omega", "f64", rpy, dt, out_shape=(3, 3)), tw.jac_omega(rpy, dt), rtol=1e-13, atol=1e-16)
        np.testing.assert_allclose(_call(lib, "orc_qtran", "f64", dt, om, out_shape=(4, 4)), tw.qtran(dt, om), atol=1e-15)
        # E * E^-1 = I
        E = _call(lib, "orc_rpy_to_ear_base", "f64", rpy, out_shape=(3, 3))
        Ei = _call(lib, "orc_rpy_to_ear_base_inv", "f64", rpy, out_shape=(3, 3))
        np.testing.assert_allclose(E @ Ei, np.eye(3), atol=1e-12)
        # the EKF Jacobian blocks are the derivatives of rpy + dt*Einv(rpy)*omega
        def f(r, w):
            return r + dt * tw.ear_base_inv(r) @ w
        h = 1e-6
        Jr = np.stack([(f(rpy + h * e, om) - f(rpy - h * e, om)) / (2 * h) for e in np.eye(3)], 1)
        Jw = np.stack([(f(rpy, om + h * e) - f(rpy, om - h * e)) / (2 * h) for e in np.eye(3)], 1)
        np.testing.assert_allclose(tw.jac_rpy(rpy, om, dt), Jr, atol=1e-6)
        np.testing.assert_allclose(tw.jac_omega(rpy, dt), Jw, atol=1e-7)


def test_unwrap_continuity():
    lib = oracle.load()
    prev = 0.0
    truth = 0.0
    for k in range(2000):
        truth += 0.05
        wrapped = tw.constrain_angle(truth)
        prev = lib.orc_unwrap_f64(prev, wrapped)
        assert prev == pytest.approx(truth, abs=1e-9)
    for a in np.linspace(-20, 20, 401):
        assert lib.orc_constrain_angle_f64(a) == pytest.approx(tw.constrain_angle(a), abs=1e-15)
        assert -np.pi <= lib.orc_constrain_angle_f64(a) < np.pi
