"""Synthetic measurement streams (SURVEY 8d / BASELINE.md configs), generated on the device by the library's
counter-based generator (csrc/stream_gen.hpp, C symbols target_stream_*): every value is a pure function of
(seed, target, tick, component), so a CPU checker regenerates the identical doubles without copying anything back
(tests: oracle.stream_fill, held to bit equality in tests/test_stream_gen.py).

Truth: p(t) = p0 + v t (+ a t^2/2 for the accelerated model), body rate omega turning the orientation as the reference's
Qtran(dt, omega) does tick after tick (test/target_manager_test.cpp:106-113), in closed form.  Measurements: xyz +
N(0, 0.01^2), noiseless quaternion unless rpy_noise > 0.  torch only allocates the buffers here: no torch kernel runs.
"""
import ctypes as C

import torch

from . import capi
from .manager import ANGULAR_RATES, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY  # noqa: F401


def make_stream(model, n_targets, ticks, dt, seed, device="cuda", availability=1.0, rpy_noise=0.0, dtype="f64",
                first_target=0, first_tick=0, stream=None):
    """Returns dict(p0 [N,7] f64 (pose to create the targets with), meas [ticks,7,N] SoA in precision `dtype` (pose at
    t=(first_tick+s+1)dt), has_meas [ticks,N] uint8 or None, v, a, omega truth [N,3] f64).  availability < 1: each
    target has a measurement on a tick with that probability (predict-only otherwise); rpy_noise > 0: the measured
    orientation is the true one rotated by a small random rotation vector of that standard deviation per axis (rad).
    Asynchronous on `stream` (a hipStream_t as int; default: torch's current stream)."""
    lib = capi.lib()
    N, T = int(n_targets), int(ticks)
    spec = capi.StreamSpec(int(model), int(seed), int(first_target), float(dt), float(availability), float(rpy_noise))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    meas = torch.empty((T, 7, N), dtype=tdt, device=device)
    has = torch.empty((T, N), dtype=torch.uint8, device=device) if availability < 1.0 else None
    p0 = torch.empty((N, 7), dtype=torch.float64, device=device)
    truth = torch.empty((N, 12), dtype=torch.float64, device=device)
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    if N and T:
        rc = lib.target_stream_fill_dev(C.byref(spec), N, int(first_tick), T, 0 if dtype == "f64" else 1, meas.data_ptr(), 7 * N, N,
                                        None if has is None else has.data_ptr(), N, st)
        if rc != 0:
            raise RuntimeError("target_stream_fill_dev failed: %s" % capi.last_error())
    if N:
        rc = lib.target_stream_truth_dev(C.byref(spec), N, p0.data_ptr(), truth.data_ptr(), st)
        if rc != 0:
            raise RuntimeError("target_stream_truth_dev failed: %s" % capi.last_error())
    return dict(p0=p0, meas=meas, has_meas=has, v=truth[:, 3:6], a=truth[:, 6:9], omega=truth[:, 9:12], p=truth[:, 0:3],
                spec=dict(model=int(model), seed=int(seed), first_target=int(first_target), first_tick=int(first_tick), dt=float(dt),
                          availability=float(availability), rpy_noise=float(rpy_noise), dtype=dtype))
