#!/usr/bin/env python3
"""Latency of the batch extension's host-array calls at node-tick sizes: one target_manager_update_meas_batch (ids + measurements from
host arrays) followed by one target_manager_get_est_batch (pose + twist + acceleration back to host arrays), for T targets.
    python tools/batch_call_latency.py          (GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import target_estimation_amd as te  # noqa: E402

for model in ("uniform_velocity", "angular_velocities"):
    # (targets in the manager, targets per call): whole-population calls, and calls that name a part of a larger population
    for P, T in ((1, 1), (40, 40), (400, 400), (4000, 4000), (4000, 400), (16000, 40), (16000, 400), (16000, 1000)):
        m = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % model))
        all_ids = np.arange(P, dtype=np.uint32) + 5
        m.init_batch(all_ids, 0.004, 0.0, np.tile([0.1, 0.2, 0.3, 0, 0, 0, 1.0], (P, 1)))
        ids = all_ids[:: max(1, P // T)][:T].copy()
        p = np.tile([0.1, 0.2, 0.3, 0, 0, 0, 1.0], (T, 1))
        reps = 400 if T <= 400 else 100
        for _ in range(20):
            m.update_batch(ids, 0.004, p)
            m.get_est_batch(ids)
        t0 = time.perf_counter()
        for _ in range(reps):
            m.update_batch(ids, 0.004, p)
        t1 = time.perf_counter()
        for _ in range(reps):
            m.update_batch(ids, 0.004, p)
            m.get_est_batch(ids)
        t2 = time.perf_counter()
        print("%-20s %5d of %5d targets: update_meas_batch %7.1f us per call; update_meas_batch + get_est_batch %7.1f us per tick" % (
            model, T, P, (t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6), flush=True)
        m.close()
