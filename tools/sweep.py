#!/usr/bin/env python3
"""Sweep lanes-per-target / batch size / precision for every model; one line per configuration.
Run on the GPU box:  python tools/sweep.py [--steps 200] [--sizes 10000,1000000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LANES = {"uniform_velocity": {"f64": [1, 3, 101, 103, 201, 301], "f32": [1, 3, 101, 103, 201, 301]},
         "uniform_acceleration": {"f64": [1, 3, 101, 103, 201, 301], "f32": [1, 3, 101, 103, 201, 301]},
         "angular_rates": {"f64": [3, 6, 103, 106, 201, 301], "f32": [2, 3, 6, 102, 103, 106, 201, 301]},
         "angular_velocities": {"f64": [3, 6, 101, 103, 106, 201, 301], "f32": [1, 3, 6, 101, 103, 106, 201, 301]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--sizes", default="10000,100000,1000000")
    ap.add_argument("--models", default="uniform_velocity,uniform_acceleration,angular_velocities,angular_rates")
    ap.add_argument("--dtypes", default="f64,f32")
    args = ap.parse_args()
    import torch
    import target_estimation_amd as te
    torch.cuda.set_device(0)
    print("%-22s %-4s %2s %9s %10s %12s %9s %6s" % ("model", "prec", "G", "targets", "us/tick", "cycles/s", "alg GB/s", "frac"))
    for model in args.models.split(","):
        for dtype in args.dtypes.split(","):
            for n in [int(s) for s in args.sizes.split(",")]:
                for g in LANES[model][dtype]:
                    bench.WORKLOADS["_sweep"] = ("sweep", model, dtype, n, 7)
                    steps = args.steps if n <= 200000 else max(20, args.steps // 5)
                    try:
                        r = bench.run_workload(te, torch, "_sweep", steps, 10, g, stream_ticks=8, reps=2)
                    except Exception as exc:
                        print("%-22s %-4s %3d %9d  ERROR %s" % (model, dtype, g, n, str(exc)[:80]), flush=True)
                        continue
                    print("%-22s %-4s %3d %9d %10.2f %12.4g %9.0f %6.3f" % (model, dtype, g, n, r["device_ms_per_launch"] * 1e3,
                          r["cycles_per_s"], r["achieved_gbs"], r["achieved_gbs"] / 8000.0), flush=True)
                    torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
