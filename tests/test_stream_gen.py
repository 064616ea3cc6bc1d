"""The counter-based synthetic stream generator (SURVEY 8d "Synthetic inputs": keyed (seed, target, tick, component), so
CPU and GPU regenerate identical streams).  Product: csrc/stream_gen.hpp + target_stream_* C symbols; checker: the
oracle's C twin oracle/te_stream.c.  What the streams restate: test/target_manager_test.cpp:82-115."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

MODELS = {"angular_rates": 0, "angular_velocities": 1, "uniform_acceleration": 2, "uniform_velocity": 3}


def test_product_generator_on_the_host_equals_the_oracle_twin_bit_for_bit(tmp_path):
    import oracle
    oracle.load()
    build = os.path.join(ROOT, "oracle", "_build")
    exe = str(tmp_path / "stream_gen_host_test")
    src = os.path.join(ROOT, "tests", "host", "stream_gen_host_test.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-Wno-unknown-pragmas", "-o", exe, src,
                           "-L", build, "-lte_oracle", "-Wl,-rpath," + build])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "stream generator host test ok" in out.stdout


def test_stream_is_what_the_reference_test_generates():
    """Line + N(0, 0.01^2) on xyz (test/target_manager_test.cpp:11-12,:102-104) and a unit quaternion turned by
    Qtran(dt, omega) every tick (:106-113), per target."""
    import oracle
    from oracle import np_twin as tw
    dt, N, T = 0.004, 20000, 6
    s = oracle.stream_fill(MODELS["uniform_acceleration"], 20240003, N, T, dt)
    tr = s["truth"]
    assert (np.abs(tr[:, 0:3]) <= 10).all() and (np.abs(tr[:, 3:6]) <= 1).all()
    assert (np.abs(tr[:, 6:8]) <= 0.1).all() and (np.abs(tr[:, 8] + 9.81) <= 0.1).all()
    assert (np.abs(tr[:, 9]) <= 3).all() and (np.abs(tr[:, 10:12]) <= 0.1).all()
    for k in range(T):
        t = (k + 1) * dt
        noise = s["meas"][k, :, :3] - (tr[:, 0:3] + tr[:, 3:6] * t + 0.5 * tr[:, 6:9] * t * t)
        assert abs(noise.mean()) < 3e-4 and noise.std() == pytest.approx(0.01, rel=0.02)
        assert np.abs(noise).max() < 0.07
    # draws of different ticks / targets / components are uncorrelated
    n0 = (s["meas"][0, :, :3] - (tr[:, 0:3] + tr[:, 3:6] * dt + 0.5 * tr[:, 6:9] * dt * dt)) / 0.01
    n1 = (s["meas"][1, :, :3] - (tr[:, 0:3] + tr[:, 3:6] * 2 * dt + 0.5 * tr[:, 6:9] * 4 * dt * dt)) / 0.01
    assert abs((n0 * n1).mean()) < 0.02 and abs((n0[:, 0] * n0[:, 1]).mean()) < 0.03 and abs((n0[:-1, 0] * n0[1:, 0]).mean()) < 0.03
    # the quaternion recurrence of the reference's generator, for a few targets
    for i in range(5):
        q = np.array([0, 0, 0, 1.0])
        M = tw.qtran(dt, tr[i, 9:12])
        for k in range(T):
            q = tw.quat_normalize(M @ q)
            np.testing.assert_allclose(s["meas"][k, i, 3:], q, atol=1e-14)
    # the other models have no acceleration; the same key gives the same start and velocity
    u = oracle.stream_fill(MODELS["uniform_velocity"], 20240003, 100, 1, dt)
    assert (u["truth"][:, 6:9] == 0).all()
    np.testing.assert_array_equal(u["truth"][:, :6], tr[:100, :6])
    np.testing.assert_array_equal(u["p0"][:, 3:], np.tile([0, 0, 0, 1.0], (100, 1)))


def test_keyed_generation_is_order_independent():
    """Any (target, tick) block equals the same block of a larger fill: shards of a multi-GPU run and rings refilled
    later see the same stream."""
    import oracle
    full = oracle.stream_fill(0, 5, 64, 12, 0.004, availability=0.7, rpy_noise=0.1)
    part = oracle.stream_fill(0, 5, 16, 4, 0.004, first_target=32, first_tick=6, availability=0.7, rpy_noise=0.1)
    np.testing.assert_array_equal(part["meas"], full["meas"][6:10, 32:48])
    np.testing.assert_array_equal(part["has_meas"], full["has_meas"][6:10, 32:48])
    np.testing.assert_array_equal(part["p0"], full["p0"][32:48])
    assert 0.6 < full["has_meas"].mean() < 0.8
    assert not np.array_equal(full["meas"], oracle.stream_fill(0, 6, 64, 12, 0.004, availability=0.7, rpy_noise=0.1)["meas"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_device_stream_equals_the_oracle_twin_bit_for_bit(name, dtype):
    import torch
    import oracle
    from target_estimation_amd.streams import make_stream
    N, T, dt = 3001, 9, 0.004
    kw = dict(availability=0.85, rpy_noise=0.1) if name.startswith("angular") else {}
    st = make_stream(MODELS[name], N, T, dt, 20240004, dtype=dtype, first_target=777, first_tick=3, **kw)
    torch.cuda.synchronize()
    ref = oracle.stream_fill(MODELS[name], 20240004, N, T, dt, first_target=777, first_tick=3, dtype=dtype, **kw)
    meas = st["meas"].to(torch.float64).cpu().numpy().transpose(0, 2, 1)     # [T, N, 7]
    np.testing.assert_array_equal(meas, ref["meas"])
    np.testing.assert_array_equal(st["p0"].cpu().numpy(), ref["p0"])
    np.testing.assert_array_equal(torch.cat([st["p"], st["v"], st["a"], st["omega"]], 1).cpu().numpy(), ref["truth"])
    if kw:
        np.testing.assert_array_equal(st["has_meas"].cpu().numpy(), ref["has_meas"])
    else:
        assert st["has_meas"] is None


@pytest.mark.gpu
def test_stream_fill_refuses_bad_arguments():
    import ctypes as C
    import torch
    from target_estimation_amd import capi
    lib = capi.lib()
    buf = torch.empty(7 * 16, dtype=torch.float64, device="cuda")
    spec = capi.StreamSpec(3, 1, 0, 0.004, 1.0, 0.0)
    assert lib.target_stream_fill_dev(C.byref(spec), 16, 0, 1, 0, buf.data_ptr(), 7 * 16, 8, None, 0, None) < 0    # ld < n
    assert lib.target_stream_fill_dev(C.byref(spec), 16, 0, 1, 7, buf.data_ptr(), 7 * 16, 16, None, 0, None) < 0   # dtype
    bad = capi.StreamSpec(9, 1, 0, 0.004, 1.0, 0.0)
    assert lib.target_stream_fill_dev(C.byref(bad), 16, 0, 1, 0, buf.data_ptr(), 7 * 16, 16, None, 0, None) < 0    # model
    assert lib.target_stream_fill_dev(C.byref(spec), 16, 0, 1, 0, None, 7 * 16, 16, None, 0, None) < 0
    assert lib.target_stream_fill_dev(C.byref(spec), 0, 0, 1, 0, buf.data_ptr(), 0, 0, None, 0, None) == 0         # empty
    assert lib.target_stream_fill_dev(C.byref(spec), 16, 0, 1, 0, buf.data_ptr(), 7 * 16, 16, None, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(buf).all()
