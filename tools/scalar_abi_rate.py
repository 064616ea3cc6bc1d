#!/usr/bin/env python3
"""Cost of the reference's per-id C ABI at the reference's scale: T targets, each tick = T x
(target_manager_update_meas + get_est_pose + get_est_twist), through ctypes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import target_estimation_amd as te  # noqa: E402

for T in (1, 3, 40, 400):
    m = te.TargetManager(os.path.join(ROOT, "models", "model_angular_velocities_params.yaml"))
    p = np.array([0.1, 0.2, 0.3, 0, 0, 0, 1.0])
    for i in range(T):
        m.init(i, 0.004, 0.0, p)
    ticks = max(20, 2000 // T)
    for _ in range(3):
        for i in range(T):
            m.update(i, 0.004, p)
        m.getTargetPose(0)
    t0 = time.perf_counter()
    for k in range(ticks):
        for i in range(T):
            m.update(i, 0.004, p)
        for i in range(T):
            m.getTargetPose(i)
            m.getTargetTwist(i)
    el = time.perf_counter() - t0
    print("%4d targets: %.1f us per tick, %.2f us per target-cycle (update + 2 getters)" % (T, el / ticks * 1e6, el / ticks / T * 1e6), flush=True)
    m.close()
