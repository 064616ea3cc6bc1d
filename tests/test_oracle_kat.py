"""Known-answer and structural checks of the CPU oracle (no GPU)."""
import numpy as np
import pytest

import oracle
from oracle import np_twin as tw


def test_uv_step1_known_answer(models):
    """Hand-derivable first step of the uniform-velocity model with the shipped YAML matrices,
    dt = 1/250, z = p0.  P-[pp], P-[pv], P-[vv], S, K are exact in double; the
    posterior entries that go through (1-K) with K~0.999 are conditioned ~1e3, so the survey's
    division-form numbers are matched to 1e-12 relative and the inverse-multiply form the
    reference code actually evaluates (src/kalman.cpp:92-94) is matched to the last bit."""
    m = models["uniform_velocity"]
    dt = 1.0 / m["frequency"]
    p0 = np.array([0.3, -0.2, 0.1, 0, 0, 0, 1.0])
    t = oracle.OracleTarget(m["model"], m["Q"], m["R"], m["P"], p0, dt)
    t.add_measurement(dt, p0)
    x, P = t.state()
    x, P = x[0], P[0]
    np.testing.assert_array_equal(x, [0.3, -0.2, 0.1, 0, 0, 0])  # innovation is exactly zero
    for ax in range(3):
        # inverse-multiply form, bit exact
        assert P[ax, ax] == 9.990010005977634e-05
        assert P[ax, ax + 3] == 3.995997611991675e-08
        assert P[ax + 3, ax] == 3.995997611991542e-08
        assert P[ax + 3, ax + 3] == 9.999984032009539e-03
        # survey's division-form numbers
        assert P[ax, ax] == pytest.approx(9.990010005978744e-05, rel=1e-12)
        assert P[ax, ax + 3] == pytest.approx(3.995997611992119e-08, rel=1e-12)
    # (I-KC)P is not numerically symmetric (SURVEY headline 2)
    assert P[0, 3] != P[3, 0]
    off = P.copy()
    for ax in range(3):
        for a in (ax, ax + 3):
            for b in (ax, ax + 3):
                off[a, b] = 0
    assert np.all(off == 0)


@pytest.mark.parametrize("name", ["uniform_velocity", "uniform_acceleration", "angular_rates",
                                  "angular_velocities"])
def test_predict_only_is_linear_propagation(models, name):
    """update(dt) leaves x = A^k x0 for the linear models and never touches n_meas."""
    m = models[name]
    n = m["Q"].shape[0]
    dt = 0.004
    rng = np.random.default_rng(5)
    p0 = np.concatenate([rng.uniform(-1, 1, 3), tw.rpy_to_quat(rng.uniform(-0.5, 0.5, 3))])
    v0 = rng.uniform(-1, 1, 6) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
    a0 = rng.uniform(-1, 1, 6) * 0.1
    t = oracle.OracleTarget(m["model"], m["Q"], m["R"], m["P"], p0, dt, v0=v0, a0=a0)
    ref = tw.Target(m["model"], m["Q"], m["R"], m["P"], p0, dt, v0=v0, a0=a0)
    x0, _ = t.state()
    for _ in range(25):
        t.update(dt)
        ref.update(dt)
    x, P = t.state()
    np.testing.assert_allclose(x[0], ref.x, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(P[0], ref.P, rtol=1e-11, atol=1e-18)
    if name != "angular_velocities":
        A = ref._A(dt)
        np.testing.assert_allclose(x[0], np.linalg.matrix_power(A, 25) @ x0[0], rtol=1e-12, atol=1e-14)
    # covariance stays symmetric PSD under pure prediction
    np.testing.assert_allclose(P[0], P[0].T, rtol=1e-10, atol=1e-20)
    assert np.linalg.eigvalsh(0.5 * (P[0] + P[0].T)).min() > -1e-15


def test_inverse_matches_numpy():
    rng = np.random.default_rng(1)
    lib = oracle.load()
    import ctypes as C
    for n in (3, 6):
        for _ in range(20):
            B = rng.normal(size=(n, n))
            S = B @ B.T + 1e-3 * np.eye(n)
            out = np.zeros((n, n))
            rc = lib.orc_inverse_f64(n, S.ctypes.data_as(C.POINTER(C.c_double)),
                                     out.ctypes.data_as(C.POINTER(C.c_double)))
            assert rc == 0
            np.testing.assert_allclose(out, np.linalg.inv(S), rtol=1e-9, atol=1e-12)
    # needs pivoting
    A = np.array([[0.0, 2.0, 1.0], [1.0, 0.0, 3.0], [4.0, 1.0, 0.0]])
    out = np.zeros((3, 3))
    assert lib.orc_inverse_f64(3, A.ctypes.data_as(C.POINTER(C.c_double)),
                               out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    np.testing.assert_allclose(out @ A, np.eye(3), atol=1e-14)


def test_lowest_real_root_semantics():
    """src/intersection_solver.cpp:4-17: -1 if the leading coefficient is 0; the smallest real
    part among roots with |imag| < 1e-10 (may be negative -- the caller maps that to -1)."""
    assert oracle.lowest_real_root([1.0, 2.0, 1.0, 0.0, 0.0]) == -1
    # (x-1)(x-2)(x-3)(x-4)
    c = np.poly([1, 2, 3, 4])[::-1]
    assert oracle.lowest_real_root(c) == pytest.approx(1.0, abs=1e-12)
    # (x+5)(x-2)(x^2+1): smallest real root is -5
    c = np.polymul(np.poly([-5, 2]), [1, 0, 1])[::-1]
    assert oracle.lowest_real_root(c) == pytest.approx(-5.0, abs=1e-12)
    # no real roots
    c = np.polymul([1, 0, 1], [1, 0, 4])[::-1]
    assert oracle.lowest_real_root(c) == -1
    rng = np.random.default_rng(3)
    for _ in range(200):
        c = rng.normal(size=5)
        mine = np.sort_complex(oracle.poly_roots(c))
        ref = np.sort_complex(np.roots(c[::-1]))
        np.testing.assert_allclose(mine, ref, rtol=1e-8, atol=1e-8)
        assert oracle.lowest_real_root(c) == pytest.approx(
            min([z.real for z in np.roots(c[::-1]) if abs(z.imag) < 1e-10], default=-1), rel=1e-9, abs=1e-9)


def test_moving_average_filter_matches_reference_test():
    """test/avg_filter_test.cpp restated: MovingAvgFilter(1000) on 10 000 N(5,1) samples ends with
    mean within 0.1 of 5 and variance within 0.1 of 1 (:30-41); plus the warm-up rule (mean over the
    samples seen so far until the window is full, utils.hpp:231-237)."""
    import ctypes as C
    lib = oracle.load()
    buf = (C.c_char * 8300)()
    f = C.c_void_p(C.addressof(buf))
    lib.orc_moving_avg_init(f, 1000)
    rng = np.random.default_rng(0)
    vals = rng.normal(5.0, 1.0, 10000)
    for i, v in enumerate(vals):
        m = lib.orc_moving_avg_update(f, float(v))
        if i < 1000:
            assert m == pytest.approx(vals[:i + 1].mean(), rel=1e-12)
    assert m == pytest.approx(vals[-1000:].mean(), rel=1e-12)
    assert abs(m - 5.0) < 0.1
    variance = C.cast(C.addressof(buf) + 24, C.POINTER(C.c_double))[0]
    assert abs(variance - 1.0) < 0.1
