#!/usr/bin/env python3
"""Experiment: the two batches of the mixed headline tick (500 000 AR + 500 000 AV, fp64) launched (A) one after the other on one
stream, as bench.py's sequence mode does, (B) concurrently on two streams with a join after every tick, (C) concurrently and
free-running (no join: the chains drift, not a tick-lockstep schedule).  Python launches; the kernels are 70-80 us long."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import target_estimation_amd as te
from target_estimation_amd.streams import make_stream

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
dt, ticks = 1.0 / 250.0, 16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
mgrs, batches, meas = [], [], []
MODELS = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("angular_rates", "angular_velocities")
for k, model in enumerate(MODELS):
    mt = te.MODEL_TYPES[model]
    st = make_stream(mt, N, ticks, dt, 77 + k)
    m = te.TargetManager(dtype=dtype)
    params = bench._model_params(model)
    m.init_batch(np.arange(N, dtype=np.uint32), dt, 0.0, st["p0"].cpu().numpy(), None, None, type=mt, Q=params["Q"], R=params["R"], P0=params["P"])
    b = m.batches()[0]
    mgrs.append(m); batches.append(b); meas.append(st["meas"].to(b.torch_dtype()).contiguous())
torch.cuda.synchronize()


def run(mode, steps):
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    if mode == "A":
        for m in mgrs: m.set_stream(s1.cuda_stream)
    else:
        mgrs[0].set_stream(s1.cuda_stream); mgrs[1].set_stream(s2.cuda_stream)
    for s in range(steps):
        order = (0, 1) if (mode != "A" or s % 2 == 0) else (1, 0)   # A: whole-tick reversal across the batches
        for j in order:
            batches[j].step(dt, meas[j][s % ticks])
        if mode == "B":
            e1.record(s1); e2.record(s2)
            s1.wait_event(e2); s2.wait_event(e1)


for mode in ("A", "B", "C", "A", "B", "C"):
    run(mode, 40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(mode, 400)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 400
    byts = sum(b.algorithmic_bytes for b in batches) * N
    print("mode %s: %.1f us per tick, %.0f GB/s" % (mode, t * 1e6, byts / t / 1e9), flush=True)
