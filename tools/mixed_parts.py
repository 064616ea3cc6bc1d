#!/usr/bin/env python3
"""Where does the tick of a small mixed population go?  The parts of BASELINE configs[4]'s per-GPU share (62 500 angular-rates +
62 500 uniform-acceleration + the sphere query, fp32) timed one by one through bench.py's own run_mixed (recorded graph, 512 ticks):
each model alone, with and without the fused query, both without it, and the same at twice the size.
    python tools/mixed_parts.py [f32|f64]          (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import target_estimation_amd as te  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
AR, AV, UA, UV = "angular_rates", "angular_velocities", "uniform_acceleration", "uniform_velocity"
n = 62_500
CASES = [
    ("AR + UA + query (cfg5)", [(AR, n), (UA, n)], True),
    ("AR + UA", [(AR, n), (UA, n)], False),
    ("AR + query", [(AR, n)], True),
    ("AR", [(AR, n)], False),
    ("UA + query", [(UA, n)], True),
    ("UA", [(UA, n)], False),
    ("AR + AV (cfg4)", [(AR, n), (AV, n)], False),
    ("AV", [(AV, n)], False),
    ("AV + query", [(AV, n)], True),
    ("2 x (AR + UA + query)", [(AR, 2 * n), (UA, 2 * n)], True),
    ("AR + query, 2 x", [(AR, 2 * n)], True),
    ("AR + UA + query (cfg5), again", [(AR, n), (UA, n)], True),
]
only = os.environ.get("TE_PARTS_ONLY")
if only:
    CASES = [c for k, c in enumerate(CASES) if str(k) in only.split(",")]
print("# tools/mixed_parts.py %s: us per tick (recorded graph of launches, 512 ticks x 3 repetitions, device time)" % dtype)
for k, (label, parts, q) in enumerate(CASES):
    name = "parts_%d" % k
    bench.MIXED[name] = (label, parts, dtype, 20240005, q)
    r = bench.run_mixed(te, torch, name, 512, 64, launch_mode="graph", reps=3)
    print("%-34s %7.2f us   (%s)" % (label, r["device_ms_per_step"] * 1e3, r.get("launch_mode", "")), flush=True)
