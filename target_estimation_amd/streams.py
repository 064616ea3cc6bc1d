"""Synthetic measurement streams, generated on the device (SURVEY 8d / BASELINE.md configs).

Truth: p(t) = p0 + v t (+ a t^2/2 for the accelerated model), body rate omega integrated with the
reference's quaternion transition Qtran(dt, omega) and renormalised every tick, exactly as the
reference's test generator does (test/target_manager_test.cpp:106-113).  Measurements: xyz +
N(0, 0.01^2), noiseless quaternion.  Everything is keyed by an integer seed, so a CPU checker can
regenerate the identical stream by copying the tensors back.
"""
import math

import torch

from .manager import ANGULAR_RATES, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY  # noqa: F401


def qtran_matrix(dt, omega):
    """Qtran(dt, omega) for a batch of body rates [N,3] -> [N,4,4]; quaternion order [x y z w]
    (reference: include/target_estimation/geometry.hpp:448-465, :493-504)."""
    n = omega.norm(dim=1)
    wx, wy, wz = omega[:, 0], omega[:, 1], omega[:, 2]
    z = torch.zeros_like(wx)
    S = 0.5 * torch.stack([torch.stack([z, -wz, wy, wx], 1), torch.stack([wz, z, -wx, wy], 1),
                           torch.stack([-wy, wx, z, wz], 1), torch.stack([-wx, -wy, -wz, z], 1)], 1)
    tmp = n * dt / 2.0
    eye = torch.eye(4, dtype=omega.dtype, device=omega.device).expand(len(n), 4, 4)
    safe = torch.where(n > 0, n, torch.ones_like(n))
    M = torch.cos(tmp)[:, None, None] * eye + (2.0 / safe * torch.sin(tmp))[:, None, None] * S
    return torch.where((n > 0)[:, None, None], M, eye)


def make_stream(model, n_targets, ticks, dt, seed, device="cuda", availability=1.0, rpy_noise=0.0):
    """Returns dict(p0 [N,7] f64 (pose at t=0), meas [ticks,7,N] f64 SoA (pose at t=(s+1)dt),
    has_meas [ticks,N] uint8 or None, v, a, omega truth).  availability < 1: each target has a measurement on a tick
    with that probability (predict-only otherwise); rpy_noise > 0: the measured orientation is the true one
    rotated by a small random rotation vector of that standard deviation per axis (rad)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f64 = dict(dtype=torch.float64, device=device)
    N = int(n_targets)
    U = lambda lo, hi, *shape: lo + (hi - lo) * torch.rand(*shape, generator=g, **f64)  # noqa: E731
    p = U(-10.0, 10.0, N, 3)
    v = U(-1.0, 1.0, N, 3)
    a = torch.zeros(N, 3, **f64)
    if model == UNIFORM_ACCELERATION:
        a = torch.tensor([0.0, 0.0, -9.81], **f64) + U(-0.1, 0.1, N, 3)
    omega = torch.stack([U(-3.0, 3.0, N), U(-0.1, 0.1, N), U(-0.1, 0.1, N)], 1)
    q = torch.zeros(N, 4, **f64)
    q[:, 3] = 1.0
    M = qtran_matrix(dt, omega)
    p0 = torch.cat([p + 0.01 * torch.randn(N, 3, generator=g, **f64), q], 1)
    meas = torch.empty(ticks, 7, N, **f64)
    for s in range(ticks):
        t = (s + 1) * dt
        q = torch.bmm(M, q[:, :, None])[:, :, 0]
        q = q / q.norm(dim=1, keepdim=True)
        pos = p + v * t + 0.5 * a * (t * t)
        meas[s, 0:3] = (pos + 0.01 * torch.randn(N, 3, generator=g, **f64)).T
        qm = q
        if rpy_noise > 0.0:
            h = 0.5 * rpy_noise * torch.randn(N, 3, generator=g, **f64)          # half rotation vector
            dq = torch.cat([h, torch.ones(N, 1, **f64)], 1)
            dq = dq / dq.norm(dim=1, keepdim=True)
            x1, y1, z1, w1 = q.unbind(1)
            x2, y2, z2, w2 = dq.unbind(1)
            qm = torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                              w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], 1)
        meas[s, 3:7] = qm.T
    has = None
    if availability < 1.0:
        has = (torch.rand(ticks, N, generator=g, device=device) < availability).to(torch.uint8)
    return dict(p0=p0, meas=meas, has_meas=has, v=v, a=a, omega=omega)
