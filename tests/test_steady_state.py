"""An oracle-independent pin of the predict/update algebra: for the linear models the posterior covariance of
`P <- A P A^T + Q;  K = P C^T (C P C^T + R)^-1;  P <- (I - K C) P` (src/kalman.cpp:84-95) converges to the fixed
point that SciPy's discrete algebraic Riccati solver gives for (A(dt), C = [I 0], Q, R) of the shipped model files.
Neither the oracle nor the kernels share any code with that solver."""
import numpy as np
import pytest
import scipy.linalg as sl

import oracle
from conftest import model_path

CASES = [("uniform_velocity", 3, 2, 20000), ("uniform_acceleration", 3, 3, 6000), ("angular_rates", 6, 3, 40000)]


def riccati_posterior(m, K, NB):
    dt = 1.0 / m["frequency"]
    n = K * NB
    A = np.eye(n)                                   # uniform_velocity.cpp:90-96, uniform_acceleration.cpp:91-99, angular_rates.cpp:108-115
    for b in range(NB - 1):
        A[b * K:(b + 1) * K, (b + 1) * K:(b + 2) * K] = dt * np.eye(K)
    if NB == 3:
        A[0:K, 2 * K:3 * K] = 0.5 * dt * dt * np.eye(K)
    C = np.zeros((K, n))
    C[:, :K] = np.eye(K)
    prior = sl.solve_discrete_are(A.T, C.T, m["Q"], m["R"])
    S = C @ prior @ C.T + m["R"]
    return prior - prior @ C.T @ np.linalg.solve(S, C @ prior), dt


@pytest.mark.parametrize("name,K,NB,steps", CASES)
def test_oracle_converges_to_the_riccati_fixed_point(models, name, K, NB, steps):
    m = models[name]
    want, dt = riccati_posterior(m, K, NB)
    p0 = np.array([[0.3, -0.2, 0.1, 0, 0, 0, 1.0]])
    ob = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt)
    for _ in range(steps):
        ob.step(dt, p0)
    _, P = ob.state()
    assert np.abs(P[0] - want).max() <= 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 201, 3])
@pytest.mark.parametrize("name,K,NB,steps", CASES)
def test_gpu_converges_to_the_riccati_fixed_point(models, name, K, NB, steps, lanes):
    torch = pytest.importorskip("torch")
    te = pytest.importorskip("target_estimation_amd")
    m = models[name]
    want, dt = riccati_posterior(m, K, NB)
    N = 70
    p0 = np.tile([0.3, -0.2, 0.1, 0, 0, 0, 1.0], (N, 1))
    ids = np.arange(N, dtype=np.uint32)
    mgr = te.TargetManager(model_path(name), dtype="f64", lanes_per_target=lanes)
    mgr.init_batch(ids, dt, 0.0, p0)
    b = mgr.batches()[0]
    block = 1000
    meas = torch.from_numpy(np.ascontiguousarray(p0.T)).cuda()[None].expand(block, 7, N).contiguous()
    for _ in range(steps // block):
        b.step_sequence(dt, meas)
    _, P = mgr.get_state_batch(ids[[0, N - 1]])
    for k in range(2):
        assert np.abs(P[k] - want).max() <= 1e-9 * np.abs(want).max()
    mgr.close()
