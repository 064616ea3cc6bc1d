#!/usr/bin/env python3
"""Resident-mode capacity (targets per batch) of every model and precision on this device: a plain session, and one with the
per-tick sphere query / pose output (the larger kernel variant).

    python tools/live_capacity.py            what the library allows (target_batch_live_capacity)
    python tools/live_capacity.py --probe    ... and, per kernel, the largest session that really STARTS (the relay's start word,
                                             Batch::live_start), found by bisection with the library's own limit overridden
                                             (TE_LIVE_CAPACITY_WAVES; each refused start costs the 2 s start timeout)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(n, d) for n in ("uniform_velocity", "uniform_acceleration", "angular_velocities", "angular_rates") for d in ("f32", "f64")]


def starts(name, dtype, waves):
    """One attempt in a child process (a refused start leaves a queued kernel behind: let it die with the process)."""
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import target_estimation_amd as te
from target_estimation_amd.streams import make_stream
name, dtype, N = %r, %r, %d
st = make_stream(te.MODEL_TYPES[name], N, 2, 0.004, 5, dtype=dtype)
m = te.TargetManager(os.path.join(%r, "models", "model_%%s_params.yaml" %% name), dtype=dtype)
m.init_batch(np.arange(N, dtype=np.uint32), 0.004, 0.0, st["p0"].cpu().numpy())
b = m.batches()[0]
try:
    b.live_start(0.004, st["meas"], None, max_ticks=2, idle_limit_s=1.0)
    b.live_post(2)
    ok = b.live_wait(2, 5.0)
    b.live_stop()
    print("STARTED" if ok else "NOTICKS")
except RuntimeError as e:
    print("REFUSED", str(e)[-80:])
os._exit(0)
""" % (ROOT, name, dtype, waves * 64, ROOT)
    env = dict(os.environ, TE_LIVE_CAPACITY_WAVES=str(waves + 1))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout
    return "STARTED" in out


def main():
    import torch
    import target_estimation_amd as te
    probe = "--probe" in sys.argv
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for name, dtype in CASES:
        m = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % name), dtype=dtype)
        p0 = np.zeros((64, 7)); p0[:, 6] = 1.0
        m.init_batch(np.arange(64, dtype=np.uint32), 0.004, 0.0, p0)
        b = m.batches()[0]
        plain = b.live_capacity
        buf = torch.zeros((7, 64), dtype=torch.float64, device="cuda")
        b.live_set_pose_output(buf)
        with_out = b.live_capacity
        b.live_set_pose_output(None)
        m.close()
        line = "%-22s %s  plain %7d   with query / pose output %7d" % (name, dtype, plain, with_out)
        if probe:
            lo, hi = plain // 64 // 2, 2 * plain // 64 + cus          # waves: lo starts (assumed), hi does not (twice the limit)
            if not starts(name, dtype, lo):
                line += "   PROBE: half the allowed size did not start"
            else:
                while hi - lo > cus // 4:
                    mid = (lo + hi) // 2
                    if starts(name, dtype, mid):
                        lo = mid
                    else:
                        hi = mid
                line += "   largest plain session that started: %d wavefronts = %.2f per CU (allowed: %d = %.2f per CU)" % (
                    lo + 1, (lo + 1) / cus, plain // 64, plain / 64 / cus)
        print(line, flush=True)


if __name__ == "__main__":
    main()
