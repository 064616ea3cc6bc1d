#!/usr/bin/env python3
"""Generate tests/golden/harness_golden.npz with the NumPy twin (oracle/np_twin.py).

The reference ships no golden vectors for this path and cannot be built here (no Eigen), so
these fixtures come from the independent NumPy restatement of the reference sources, run on
the reference integration test's exact measurement stream (oracle/ref_stream.cpp).  They
cross-pin the C oracle and the HIP path; they are NOT reference outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from oracle import np_twin as tw  # noqa: E402
from conftest import HARNESS_ORDER, model_path  # noqa: E402

CHECKPOINTS = list(range(1, 17)) + [100, 1000, 10000]


def main():
    stream = oracle.ref_test_stream()
    out = {"checkpoints": np.array(CHECKPOINTS), "stream_head": stream[:, :16].copy(),
           "stream_sum": stream.sum(axis=1)}
    for k, name in enumerate(HARNESS_ORDER):
        m = oracle.load_model_yaml(model_path(name))
        dt = 1.0 / m["frequency"]
        t = tw.Target(m["model"], m["Q"], m["R"], m["P"], stream[k][0], dt)
        xs, Ps, poses, twists = [], [], [], []
        for i in range(max(CHECKPOINTS)):
            t.add_measurement(dt, stream[k][i])
            if i + 1 in CHECKPOINTS:
                xs.append(t.x.copy()); Ps.append(t.P.copy())
                poses.append(t.pose()); twists.append(t.twist.copy())
        out[name + "_x"] = np.array(xs)
        out[name + "_P"] = np.array(Ps)
        out[name + "_pose"] = np.array(poses)
        out[name + "_twist"] = np.array(twists)
        # a predict-only tail: 5 x update(dt) after the last checkpoint
        for _ in range(5):
            t.update(dt)
        out[name + "_x_pred5"] = t.x.copy()
        out[name + "_P_pred5"] = t.P.copy()
        out[name + "_pose_at"] = t.pose_at(t.t + 0.25)
        out[name + "_twist_at"] = t.twist_at(t.t + 0.25)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"), **out)
    print("wrote harness_golden.npz", {k: v.shape for k, v in out.items() if k.endswith("_P")})


if __name__ == "__main__":
    main()
