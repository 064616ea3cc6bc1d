#!/usr/bin/env python3
"""Resident-mode capacity (targets per batch) of every model and precision on this device: a plain session, and one with the
per-tick sphere query / pose output (the larger kernel variant)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import target_estimation_amd as te  # noqa: E402

for name in ("uniform_velocity", "uniform_acceleration", "angular_velocities", "angular_rates"):
    for dtype in ("f32", "f64"):
        m = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % name), dtype=dtype)
        p0 = np.zeros((64, 7)); p0[:, 6] = 1.0
        m.init_batch(np.arange(64, dtype=np.uint32), 0.004, 0.0, p0)
        b = m.batches()[0]
        plain = b.live_capacity
        buf = torch.zeros((7, 64), dtype=torch.float64, device="cuda")
        b.live_set_pose_output(buf)
        with_out = b.live_capacity
        b.live_set_pose_output(None)
        print("%-22s %s  plain %7d   with query / pose output %7d" % (name, dtype, plain, with_out), flush=True)
        m.close()
