"""C oracle against the committed fixtures (tests/golden/harness_golden.npz, produced by the
NumPy twin on the reference test's stream; see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import oracle
from conftest import HARNESS_ORDER, ROOT

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"))


def test_stream_matches_fixture(harness_stream):
    np.testing.assert_array_equal(harness_stream[:, :16], GOLD["stream_head"])
    np.testing.assert_allclose(harness_stream.sum(axis=1), GOLD["stream_sum"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", HARNESS_ORDER)
def test_oracle_matches_golden(models, harness_stream, name):
    k = HARNESS_ORDER.index(name)
    m = models[name]
    dt = 1.0 / m["frequency"]
    meas = harness_stream[k]
    cps = list(GOLD["checkpoints"])
    t = oracle.OracleTarget(m["model"], m["Q"], m["R"], m["P"], meas[0], dt)
    j = 0
    for i in range(max(cps)):
        t.add_measurement(dt, meas[i])
        if i + 1 == cps[j]:
            x, P = t.state()
            scale = np.abs(GOLD[name + "_P"][j]).max()
            np.testing.assert_allclose(x[0], GOLD[name + "_x"][j], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(P[0], GOLD[name + "_P"][j], rtol=1e-7, atol=1e-9 * scale)
            np.testing.assert_allclose(t.pose()[0], GOLD[name + "_pose"][j], atol=1e-9)
            np.testing.assert_allclose(t.twist()[0], GOLD[name + "_twist"][j], rtol=1e-7, atol=1e-9)
            j += 1
    for _ in range(5):
        t.update(dt)
    x, P = t.state()
    np.testing.assert_allclose(x[0], GOLD[name + "_x_pred5"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(P[0], GOLD[name + "_P_pred5"], rtol=1e-7, atol=1e-12)
    tq = 10005 * dt + 0.25
    np.testing.assert_allclose(t.pose_at(tq)[0], GOLD[name + "_pose_at"], atol=1e-8)
    np.testing.assert_allclose(t.twist_at(tq)[0], GOLD[name + "_twist_at"], rtol=1e-7, atol=1e-9)
