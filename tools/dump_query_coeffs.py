#!/usr/bin/env python3
"""The quartics of configs[4]'s per-GPU share as the fused query meets them (bench.py's cfg5 population after 16 ticks): coefficient
rows [c0..c4] of |p + v x + a x^2 / 2|^2 - r^2 per target, written to gpurun_out/query_coeffs_<dtype>.npy for tools/quartic_bench.hip and
the host-side iteration counts.     python tools/dump_query_coeffs.py [f32|f64]          (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import target_estimation_amd as te  # noqa: E402
from target_estimation_amd.streams import make_stream  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
n, dt, ticks = 62_500, 1.0 / 250.0, 16
rows = []
mgr = te.TargetManager(dtype=dtype)
meas, base = [], 0
for k, model in enumerate(["angular_rates", "uniform_acceleration"]):
    mt = te.MODEL_TYPES[model]
    st = make_stream(mt, n, ticks, dt, 20240005 + 17 * k, dtype=dtype)
    params = bench._model_params(model)
    mgr.init_batch(np.arange(n, dtype=np.uint32) + base, dt, 0.0, st["p0"].cpu().numpy(), None, None, type=mt, Q=params["Q"], R=params["R"], P0=params["P"])
    base += n
    meas.append(st["meas"])
mgr.step_sequence_all(dt, meas, use_graph=0, n_ticks=ticks)
for b in mgr.batches():
    pose, twist, acc = b.get_est()
    p = pose.cpu().numpy()[:, :3].astype(np.float64)
    v = twist.cpu().numpy()[:, :3].astype(np.float64)
    a = acc.cpu().numpy()[:, :3].astype(np.float64)
    c = np.stack([(p * p).sum(1) - 1.0, 2 * (p * v).sum(1), (v * v).sum(1) + (p * a).sum(1), (v * a).sum(1), 0.25 * (a * a).sum(1)], axis=1)
    rows.append(c)
    print(b.size, "targets: |p| median %.3g, |v| median %.3g, |a| median %.3g" % (np.median(np.linalg.norm(p, axis=1)), np.median(np.linalg.norm(v, axis=1)), np.median(np.linalg.norm(a, axis=1))))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", "query_coeffs_%s.npy" % dtype)
np.save(out, np.concatenate(rows))
print("wrote", out)
mgr.close()
