#!/usr/bin/env python3
"""Parity under extreme inputs (GPU box): positions up to 1e5, P0 scaled by 1e-4..1e4, tick lengths from 1e-4 to 0.5 s,
random masks, unnormalised quaternions of either sign, body rates up to 40 rad/s (unwrapped angles reach hundreds
of radians).  The EKF model is kept away from pitch = +-pi/2, where the reference's Euler-angle Jacobians are
singular (1/cos^2 pitch) and any two implementations diverge.  usage: python tests/extended/torture.py   (from the repo root; test infrastructure: it uses the oracle)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle, conftest
import target_estimation_amd as te
from oracle import np_twin as tw
rng = np.random.default_rng(123)
worst = {}
for trial in range(24):
    name = ["angular_rates", "angular_velocities", "uniform_acceleration", "uniform_velocity"][trial % 4]
    dtype = "f64"
    m = oracle.load_model_yaml(conftest.model_path(name))
    N, steps = 64, 600
    scale = 10.0 ** rng.uniform(-3, 5)
    p0 = np.concatenate([rng.normal(0, scale, (N, 3)), np.tile([0, 0, 0, 1.0], (N, 1))], 1)
    ekf = name == "angular_velocities"
    omega = rng.normal(0, 1, (N, 3)) * np.array([3.0 if ekf else 40.0, 0.02 if ekf else 5.0, 0.02 if ekf else 5.0]) * rng.uniform(0, 1)
    q = np.tile([0, 0, 0, 1.0], (N, 1)); q += rng.normal(0, 0.05 if ekf else 0.3, (N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    p0[:, 3:] = q
    Pscale = 10.0 ** rng.uniform(-4, 4)
    P0 = m["P"] * Pscale
    mgr = te.TargetManager(dtype=dtype, lanes_per_target=int(rng.choice([0, 201, 3 if name.startswith("uniform") else 6])))
    ids = np.arange(N, dtype=np.uint32)
    mgr.init_batch(ids, 0.004, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=P0)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], P0, p0, 0.004, dtype=dtype)
    b = mgr.batches()[0]
    v = rng.normal(0, scale * 0.1, (N, 3))
    pos = p0[:, :3].copy(); t = 0.0
    for s in range(steps):
        dt = float(rng.choice([1e-4, 0.004, 0.05] if ekf else [1e-4, 0.004, 0.05, 0.5]))
        t += dt
        pos = pos + v * dt
        M = np.stack([tw.qtran(dt, omega[i]) for i in range(N)])
        q = np.einsum("nij,nj->ni", M, q); q /= np.linalg.norm(q, axis=1, keepdims=True)
        qm = q * rng.uniform(0.5, 2.0, (N, 1)) * (1 if rng.random() < 0.5 else -1)    # unnormalised, either sign
        meas = np.concatenate([pos + rng.normal(0, 0.01, (N, 3)), qm], 1)
        mask = (rng.random(N) < 0.85).astype(np.uint8)
        tm = torch.from_numpy(np.ascontiguousarray(meas.T)).cuda()
        b.step(dt, tm, torch.from_numpy(mask).cuda())
        orc.step(dt, meas, mask)
    x, P = mgr.get_state_batch(ids)
    xo, Po = orc.state()
    ex = (np.abs(x - xo) / (1e-9 + np.abs(xo))).max()
    eP = (np.abs(P - Po) / np.abs(Po).max(axis=(1, 2), keepdims=True)).max()
    pose, twist, acc, _ = mgr.get_est_batch(ids)
    eo = max(np.abs(pose - orc.pose()).max(), np.abs(twist - orc.twist()).max())
    print("%-22s lanes %3d scale %.1e Pscale %.1e max|angle| %.0f rad: rel dx %.2e  dP/scale %.2e  outputs %.2e" % (
        name, b.lanes_per_target if b.layout == "full" else {"axis_separable": 201, "axis_separable_packed": 301}.get(b.layout, 0), scale, Pscale,
        np.abs(xo[:, 3:6]).max() if name.startswith("angular") else 0.0, ex, eP, eo), flush=True)
    assert np.isfinite(x).all() and np.isfinite(P).all()
    assert ex < 1e-6 and eP < 1e-9, "parity lost"
    mgr.close()
