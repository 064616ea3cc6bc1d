#!/usr/bin/env python3
"""Long-horizon parity of the layouts against the oracle: worst |dx| and |dP|/max|P| after many ticks.
usage: python tests/extended/drift.py [steps]   (on the GPU box; test infrastructure: it uses the oracle)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle
import target_estimation_amd as te
from conftest import synth_stream

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N = 64
for name in ("uniform_velocity", "uniform_acceleration", "angular_rates", "angular_velocities"):
    m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_%s_params.yaml" % name))
    dt = 1.0 / m["frequency"]
    p0, meas = synth_stream(name, N, steps, seed=11)
    ids = np.arange(N, dtype=np.uint32)
    for dtype in ("f64", "f32"):
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, 0.0, None, None, dtype=dtype)
        mgrs = {}
        for lanes in (201, 301, 1 if name.startswith("uniform") else 6):
            mg = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % name), dtype=dtype, lanes_per_target=lanes)
            mg.init_batch(ids, dt, 0.0, p0)
            mgrs[lanes] = mg
        for s in range(steps):
            orc.step(dt, meas[s])
            for lanes, mg in mgrs.items():
                b = mg.batches()[0]
                t = torch.from_numpy(np.ascontiguousarray(meas[s].T)).to("cuda").to(b.torch_dtype()).contiguous()
                b.step(dt, t)
        xo, Po = orc.state()
        scale = np.abs(Po).max(axis=(1, 2), keepdims=True)
        for lanes, mg in mgrs.items():
            x, P = mg.get_state_batch(ids)
            asym = np.abs(P - P.transpose(0, 2, 1)).max() / scale.max()
            print("%-22s %s lanes %3d after %d steps: |dx| %.3e  |dP|/scale %.3e  asym(P) %.1e  asym(oracle P) %.1e" % (
                name, dtype, lanes, steps, np.abs(x - xo).max(), (np.abs(P - Po) / scale).max(), asym,
                np.abs(Po - Po.transpose(0, 2, 1)).max() / scale.max()), flush=True)
            mg.close()
