import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_FILES = {
    "uniform_velocity": "model_uniform_velocity_params.yaml",
    "uniform_acceleration": "model_uniform_acceleration_params.yaml",
    "angular_rates": "model_angular_rates_params.yaml",
    "angular_velocities": "model_angular_velocities_params.yaml",
}
# order of the TESTs in the reference's integration test (test/target_manager_test.cpp:148-289)
HARNESS_ORDER = ["uniform_velocity", "uniform_acceleration", "angular_rates", "angular_velocities"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The HIP library and the oracle are built in-tree (and travel with the snapshot); build them on
    demand if a checkout arrives without them (hipcc / gcc are in the image)."""
    from target_estimation_amd import _build
    if not os.path.exists(_build.LIB):
        _build.build()
    import oracle
    oracle.build()


def model_path(name):
    return os.path.join(ROOT, "models", MODEL_FILES[name])


@pytest.fixture(scope="session")
def models():
    import oracle
    return {k: oracle.load_model_yaml(model_path(k)) for k in MODEL_FILES}


@pytest.fixture(scope="session")
def harness_stream():
    """The four measurement streams of test/target_manager_test.cpp (10 000 x 7 each)."""
    import oracle
    return oracle.ref_test_stream()


def synth_stream(model, N, steps, seed, dt=0.004, rpy_noise=0.0, dtype=np.float64):
    """Small seeded multi-target stream for parity tests: constant velocity (+gravity-like
    acceleration for UA) plus a body rate integrated with Qtran, xyz noise sigma 1 cm.
    Returns p0 [N,7] and meas [steps,N,7]."""
    from oracle import np_twin as tw
    rng = np.random.default_rng(seed)
    p = rng.uniform(-10, 10, (N, 3))
    v = rng.uniform(-1, 1, (N, 3))
    a = np.zeros((N, 3))
    if model in ("uniform_acceleration",):
        a = np.array([0, 0, -9.81]) + rng.uniform(-0.1, 0.1, (N, 3))
    omega = np.stack([rng.uniform(-3, 3, N), rng.uniform(-0.1, 0.1, N), rng.uniform(-0.1, 0.1, N)], 1)
    q = np.tile(np.array([0, 0, 0, 1.0]), (N, 1))
    meas = np.zeros((steps + 1, N, 7))
    for s in range(steps + 1):
        t = s * dt
        pos = p + v * t + 0.5 * a * t * t
        meas[s, :, :3] = pos + rng.normal(0, 0.01, (N, 3))
        meas[s, :, 3:] = q
        if rpy_noise > 0:
            for i in range(N):
                rpy = tw.quat_to_rpy(q[i]) + rng.normal(0, rpy_noise, 3)
                meas[s, i, 3:] = tw.rpy_to_quat(rpy)
        for i in range(N):
            q[i] = tw.quat_normalize(tw.qtran(dt, omega[i]) @ q[i])
    return meas[0].copy(), meas[1:].copy()
