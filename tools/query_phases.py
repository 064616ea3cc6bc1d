#!/usr/bin/env python3
"""Phases of the fused sphere query inside the step kernel, from the shader clock (an instrumented build of the library:
`make OUT=../lib_ts OBJ=../lib_ts/obj HIPCC="hipcc -DTE_QUERY_PHASE_CLOCK"`, loaded through TARGET_ESTIMATION_AMD_LIB).  In that build lanes 0..3 of
every wavefront return, in place of their intersection time: cycles from the wavefront's first instruction to the head of the query,
the query's coefficients, the quartic, the pose at the crossing.
    TARGET_ESTIMATION_AMD_LIB=$PWD/target_estimation_amd/lib_ts/libtarget_estimation_amd.so python tools/query_phases.py [f32|f64]"""
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
n, dt, ticks = 62_500, 1.0 / 250.0, 32
for models in (["angular_rates", "uniform_acceleration"], ["angular_rates"], ["uniform_acceleration"]):
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    meas, base = [], 0
    for k, model in enumerate(models):
        mt = te.MODEL_TYPES[model]
        st = make_stream(mt, n, ticks, dt, 20240005 + 17 * k, dtype=dtype)
        params = bench._model_params(model)
        mgr.init_batch(np.arange(n, dtype=np.uint32) + base, dt, 0.0, st["p0"].cpu().numpy(), None, None, type=mt, Q=params["Q"], R=params["R"], P0=params["P"])
        base += n
        meas.append(st["meas"])
    batches = mgr.batches()
    outs = [(torch.empty(b.size, dtype=torch.float64, device="cuda"), torch.empty((b.size, 7), dtype=torch.float64, device="cuda")) for b in batches]
    query = (np.zeros(3), 1.0, [o[0] for o in outs], [o[1] for o in outs])
    mgr.step_sequence_all(dt, meas, query=query, use_graph=0, n_ticks=ticks)
    torch.cuda.synchronize()
    print("%s, %s: shader-clock cycles, mean over the wavefronts (max)" % (" + ".join(models), dtype))
    for model, o in zip(models, outs):
        d = o[0].cpu().numpy()
        full = (len(d) // 64) * 64
        w = d[:full].reshape(-1, 64)
        names = ["start -> query", "coefficients", "quartic", "pose at the crossing"]
        print("  %-22s " % model + "   ".join("%s %6.0f (%6.0f)" % (nm, w[:, i].mean(), w[:, i].max()) for i, nm in enumerate(names)))
        hit = (w[:, 4:] > -1).mean()
        print("  %-22s lanes with a crossing: %.1f %%" % ("", 100 * hit))
    mgr.close()
