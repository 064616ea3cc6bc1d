#!/usr/bin/env python3
"""bench.py -- predict+update cycles/s of the batched Kalman path on MI355X.

One "step" = one tick = one pass of the hot path (predict + measurement update of EVERY target of the
workload) over one batch of synthetic measurements already resident in HBM.

Headline workload (N = 1): BASELINE.json configs[3] -- 1 000 000 targets, 500 000 angular-rates + 500 000
angular-velocities -- the WHOLE population on one GPU, in the reference's arithmetic (fp64,
/root/reference/src/kalman.cpp:84-95 computes in MatrixXd).  It is the largest single-GPU configuration of
BASELINE.json; configs[1] / configs[2] (10^4 / 10^5 targets, launch-bound) and everything else are reported under
`extra`.  With N > 1 every rank owns its own 10^6-target shard (weak scaling, no data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--reps R]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process touches the GPU); under the driver's own
torch.distributed.run launch the ranks are already there.  Either way the line reports the number of ranks the
process group actually had and the run fails if that differs from --gpus.

Timing: W untimed warm-up steps, then R repetitions of EXACTLY K steps, every repetition bracketed by a barrier +
torch.cuda.synchronize() on both sides, MAX over ranks per repetition; `ms_per_step` is the median repetition / K
(R is chosen so that the repetitions add up to >= 30 ms: a 20-step region of a 10^4-target batch is 0.1 ms).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured float4 copy)
L3_BYTES = 256 << 20    # Infinity Cache: a working set below this is (partly) served on-die between ticks
SURVEY_WORDS = {"uniform_velocity": 91, "uniform_acceleration": 187, "angular_velocities": 325, "angular_rates": 697}  # SURVEY 8d, full P
MODEL_N = {"uniform_velocity": 6, "uniform_acceleration": 9, "angular_velocities": 12, "angular_rates": 18}
MODEL_STRUCT = {"uniform_velocity": "ModelUV", "uniform_acceleration": "ModelUA", "angular_velocities": "ModelAV", "angular_rates": "ModelAR"}

# lanes code per workload: 0 = automatic (the shipped models are axis-separable and symmetric -> 301); *_s201 force the
# axis-separable layout with full group blocks, *_full / *_packed the dense kernel (what general matrices get).
# None = "whatever the library picks for coupled matrices" (resolved in run_workload through forced_general).
TUNED_LANES = {"cfg2_full": 3, "uv1m_full": 1, "ua1m_full": 3, "ar1m_full": 6, "av1m_full": 3, "uv1m_packed": 101,
               "ar1m_packed": 103, "av1m_packed": 101, "ar1m64_packed": 106, "av1m64_packed": 101, "ar1m64_full": 6, "av1m64_full": 6,
               "uv1m_s201": 201, "ua1m_s201": 201, "av1m_s201": 201, "ar1m_s201": 201}

GRAPH_TICKS = 64     # ticks per recorded hipGraph block (small, launch-bound batches)
GRAPH_PASSES = 4     # passes over a one-block ring per recorded graph (small batches)
SMALL = 200_000      # up to this many targets per GPU a tick is launch-bound: graphs; above: plain launches
RING_BYTES = 3 << 30  # cap of the measurement ring of a large workload
RINGS = {"cfg2_stream": 2048, "cfg3_stream": 512}
VARIANTS = {"ar1m_a90": dict(availability=0.9, rpy_noise=0.1), "av1m_a90": dict(availability=0.9, rpy_noise=0.1)}
# distinct (Q, R, P0) sets among the targets of ONE batch; *_rand: every target draws its class at random (a wavefront's
# 64 lanes then read 64 different table rows), otherwise targets of a class are neighbours (sorted class indices)
CLASSES = {"ar1m64_1kcls": 1000, "ar1m64_1kcls_rand": 1000, "ar100k64_1kcls": 1000, "uv1m_1kcls": 1000}

WORKLOADS = {
    # name: (description, model, dtype, targets per GPU, seed)
    "cfg2": ("10000 targets, uniform-velocity model, fp64 (BASELINE.json configs[1])", "uniform_velocity", "f64", 10_000, 20240002),
    "cfg3": ("100000 targets, uniform-acceleration model, fp32 (configs[2])", "uniform_acceleration", "f32", 100_000, 20240003),
    "cfg2_stream": ("configs[1] with a measurement ring of 2048 ticks (1.1 GB): every tick's measurements come from HBM, not from L2",
                    "uniform_velocity", "f64", 10_000, 20240002),
    "cfg3_stream": ("configs[2] with a measurement ring of 512 ticks (1.4 GB)", "uniform_acceleration", "f32", 100_000, 20240003),
    "uv1m": ("1000000 targets, uniform-velocity model, fp64", "uniform_velocity", "f64", 1_000_000, 20240012),
    "uv1m32": ("1000000 targets, uniform-velocity model, fp32", "uniform_velocity", "f32", 1_000_000, 20240020),
    "ua1m": ("1000000 targets, uniform-acceleration model, fp32", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "ua1m64": ("1000000 targets, uniform-acceleration model, fp64", "uniform_acceleration", "f64", 1_000_000, 20240021),
    "ar1m": ("1000000 targets, angular-rates model, fp32", "angular_rates", "f32", 1_000_000, 20240014),
    "ar1m64": ("1000000 targets, angular-rates model, fp64", "angular_rates", "f64", 1_000_000, 20240016),
    "av1m": ("1000000 targets, angular-velocities model, fp32", "angular_velocities", "f32", 1_000_000, 20240015),
    "av1m64": ("1000000 targets, angular-velocities model, fp64", "angular_velocities", "f64", 1_000_000, 20240017),
    # state beyond the 256 MB Infinity Cache (> 1 GB): the HBM-bound figures
    "uv10m": ("10000000 targets, uniform-velocity model, fp64 (1.2 GB of state)", "uniform_velocity", "f64", 10_000_000, 20240032),
    "ua10m": ("10000000 targets, uniform-acceleration model, fp64 (2.2 GB of state)", "uniform_acceleration", "f64", 10_000_000, 20240033),
    "ar4m64": ("4000000 targets, angular-rates model, fp64 (1.8 GB of state)", "angular_rates", "f64", 4_000_000, 20240036),
    "av4m64": ("4000000 targets, angular-velocities model, fp64 (1.4 GB of state)", "angular_velocities", "f64", 4_000_000, 20240037),
    "ar8m": ("8000000 targets, angular-rates model, fp32 (1.8 GB of state)", "angular_rates", "f32", 8_000_000, 20240034),
    "av8m": ("8000000 targets, angular-velocities model, fp32 (1.4 GB of state)", "angular_velocities", "f32", 8_000_000, 20240035),
    # per-target model parameters: 1000 distinct (Q, R, P0) classes in one batch (one launch per tick)
    "ar1m64_1kcls": ("1000000 targets, angular-rates model, fp64, 1000 distinct (Q, R, P0) classes in ONE batch", "angular_rates", "f64", 1_000_000, 20240041),
    "ar1m64_1kcls_rand": ("1000000 targets, angular-rates model, fp64, 1000 distinct (Q, R, P0) classes in ONE batch, classes drawn at random per target", "angular_rates", "f64", 1_000_000, 20240041),
    "ar100k64_1kcls": ("100000 targets, angular-rates model, fp64, 1000 distinct (Q, R, P0) classes in ONE batch", "angular_rates", "f64", 100_000, 20240042),
    "uv1m_1kcls": ("1000000 targets, uniform-velocity model, fp64, 1000 distinct (Q, R, P0) classes in ONE batch", "uniform_velocity", "f64", 1_000_000, 20240043),
    # forced layouts
    "cfg2_full": ("10000 targets, uniform-velocity model, fp64, dense kernel with full P", "uniform_velocity", "f64", 10_000, 20240002),
    "uv1m_full": ("1000000 targets, uniform-velocity model, fp64, dense kernel with full P", "uniform_velocity", "f64", 1_000_000, 20240012),
    "uv1m_packed": ("1000000 targets, uniform-velocity model, fp64, dense kernel with symmetric-packed P", "uniform_velocity", "f64", 1_000_000, 20240012),
    "ua1m_full": ("1000000 targets, uniform-acceleration model, fp32, dense kernel with full P", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "ar1m_full": ("1000000 targets, angular-rates model, fp32, dense kernel with full P", "angular_rates", "f32", 1_000_000, 20240014),
    "av1m_full": ("1000000 targets, angular-velocities model, fp32, dense kernel with full P", "angular_velocities", "f32", 1_000_000, 20240015),
    "ar1m_packed": ("1000000 targets, angular-rates model, fp32, dense kernel with symmetric-packed P (automatic for coupled symmetric matrices)", "angular_rates", "f32", 1_000_000, 20240014),
    "av1m_packed": ("1000000 targets, angular-velocities model, fp32, dense kernel with symmetric-packed P (automatic for coupled symmetric matrices)", "angular_velocities", "f32", 1_000_000, 20240015),
    "ar1m64_packed": ("1000000 targets, angular-rates model, fp64, dense kernel with symmetric-packed P (automatic for coupled symmetric matrices)", "angular_rates", "f64", 1_000_000, 20240016),
    "av1m64_packed": ("1000000 targets, angular-velocities model, fp64, dense kernel with symmetric-packed P (automatic for coupled symmetric matrices)", "angular_velocities", "f64", 1_000_000, 20240017),
    "ar1m64_full": ("1000000 targets, angular-rates model, fp64, dense kernel with full P (automatic for non-symmetric matrices)", "angular_rates", "f64", 1_000_000, 20240016),
    "av1m64_full": ("1000000 targets, angular-velocities model, fp64, dense kernel with full P (automatic for non-symmetric matrices)", "angular_velocities", "f64", 1_000_000, 20240017),
    "uv1m_s201": ("1000000 targets, uniform-velocity model, fp64, axis-separable layout with full group blocks", "uniform_velocity", "f64", 1_000_000, 20240012),
    "ua1m_s201": ("1000000 targets, uniform-acceleration model, fp32, axis-separable layout with full group blocks", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "av1m_s201": ("1000000 targets, angular-velocities model, fp32, axis-separable layout with full group blocks", "angular_velocities", "f32", 1_000_000, 20240015),
    "ar1m_s201": ("1000000 targets, angular-rates model, fp32, axis-separable layout with full group blocks", "angular_rates", "f32", 1_000_000, 20240014),
    "ar1m_a90": ("1000000 targets, angular-rates model, fp32, measurements on 90 % of the (target, tick) pairs, orientation noise 0.1 rad (SURVEY 8d variant)", "angular_rates", "f32", 1_000_000, 20240018),
    "av1m_a90": ("1000000 targets, angular-velocities model, fp32, measurements on 90 % of the (target, tick) pairs, orientation noise 0.1 rad (SURVEY 8d variant)", "angular_velocities", "f32", 1_000_000, 20240019),
}

# BASELINE.json configs[3] / configs[4]: mixed populations = two batches (two motion models) in one manager.
#   name: (description, [(model, targets per GPU)], dtype, seed, per-tick sphere query)
MIXED = {
    "cfg4_1gpu": ("configs[3]: 1000000 targets, 500000 angular-rates + 500000 angular-velocities, fp64 (the reference's arithmetic), the whole population on ONE GPU",
                  [("angular_rates", 500_000), ("angular_velocities", 500_000)], "f64", 20240004, False),
    "cfg4_1gpu32": ("configs[3]: 1000000 targets, 500000 angular-rates + 500000 angular-velocities, fp32, the whole population on ONE GPU",
                    [("angular_rates", 500_000), ("angular_velocities", 500_000)], "f32", 20240004, False),
    "cfg4_4m": ("configs[3] x 4: 2000000 angular-rates + 2000000 angular-velocities, fp64 (3.2 GB of state: beyond the Infinity Cache)",
                [("angular_rates", 2_000_000), ("angular_velocities", 2_000_000)], "f64", 20240004, False),
    "cfg4": ("configs[3], one GPU's share of 8: 62500 angular-rates + 62500 angular-velocities, fp32",
             [("angular_rates", 62_500), ("angular_velocities", 62_500)], "f32", 20240004, False),
    "cfg4_64": ("configs[3], one GPU's share of 8: 62500 angular-rates + 62500 angular-velocities, fp64",
                [("angular_rates", 62_500), ("angular_velocities", 62_500)], "f64", 20240004, False),
    "cfg5": ("configs[4], one GPU's share of 8: 62500 angular-rates + 62500 uniform-acceleration + sphere intersection of every target every tick, fp32",
             [("angular_rates", 62_500), ("uniform_acceleration", 62_500)], "f32", 20240005, True),
    "cfg5_1gpu": ("configs[4]: 1000000 targets, 500000 angular-rates + 500000 uniform-acceleration + sphere intersection of every target every tick, fp32, on ONE GPU",
                  [("angular_rates", 500_000), ("uniform_acceleration", 500_000)], "f32", 20240005, True),
    "cfg5_1gpu64": ("configs[4] in fp64: 500000 angular-rates + 500000 uniform-acceleration + sphere intersection every tick, on ONE GPU",
                    [("angular_rates", 500_000), ("uniform_acceleration", 500_000)], "f64", 20240005, True),
}
HEADLINE = "cfg4_1gpu"

DEFAULT_EXTRA = ("cfg2,cfg2_live,cfg2_stream,cfg3,cfg3_live,cfg3_stream,cfg4,cfg4_live,cfg4_64,cfg4_64_live,cfg5,cfg5_live,cfg4_1gpu32,cfg5_1gpu,cfg5_1gpu64,"
                 "uv1m,uv1m32,ua1m64,ua1m,av1m64,av1m,ar1m64,ar1m,"
                 "uv10m,ua10m,av4m64,ar4m64,av8m,ar8m,cfg4_4m,"
                 "ar1m_a90,av1m_a90,ar1m64_1kcls,ar1m64_1kcls_rand,ar100k64_1kcls,uv1m_1kcls,uv1m_full,uv1m_packed,ar1m_full,ar1m_packed,av1m_packed,ar1m64_full,av1m64_full,ar1m64_packed,av1m64_packed")
# N > 1: BASELINE configs[3] / configs[4] as they are stated -- 10^6 targets OVER the N GPUs (strong scaling; at N = 8 these are the
# 62 500 + 62 500 shares) -- come first, so that a deadline cuts the per-model rows, not them
DEFAULT_EXTRA_MULTI = "cfg4_1gpu_strong,cfg4_1gpu32_strong,cfg5_1gpu_strong,uv1m_strong,ar1m64_strong,uv1m,ua1m64,av1m64,ar1m64"
# On request only (--extra cfg4_1gpu_replay,cfg4_4m_replay): the mixed populations with the batches' chains free-running inside graph
# blocks.  Not in the default line: they launch the headline's kernels at the headline's grid CONCURRENTLY, which would mix overlapped
# durations into the per-kernel averages of the rocprofv3 summary that goes with the default command (profiles/r02_mixed_replay_trace.txt).


# ------------------------------------------------------------------------------------------------ helpers
def _model_params(model):
    """Q, R, P0 of a shipped model file (row-major)."""
    import numpy as np
    import yaml
    with open(os.path.join(ROOT, "models", "model_%s_params.yaml" % model)) as f:
        node = yaml.safe_load(f)
    n = MODEL_N[model]
    m = 3 if n in (6, 9) else 6
    return dict(Q=np.array(node["Q"]).reshape(n, n), R=np.array(node["R"]).reshape(m, m), P=np.array(node["P"]).reshape(n, n))


def kernel_name(batch, model):
    """The step kernel a batch launches, as rocprofv3 prints it (kf_model_*.hip instantiations)."""
    T = "double" if batch.dtype == "f64" else "float"
    lay = batch.layout
    if lay.startswith("axis_separable"):
        return "kf_step_sep_kernel<%s,%s,%d>" % (MODEL_STRUCT[model], T, 3 if lay.endswith("packed") else 2)
    return "kf_step_kernel<%s,%s,%d,%d>" % (MODEL_STRUCT[model], T, batch.lanes_per_target, 1 if lay == "symmetric_packed" else 0)


def block_ticks(steps):
    """Ticks per recorded graph block: the largest divisor of `steps` in [16, GRAPH_TICKS], else min(steps, GRAPH_TICKS)."""
    if steps % GRAPH_TICKS == 0:
        return GRAPH_TICKS
    return next((g for g in range(min(GRAPH_TICKS, steps), 15, -1) if steps % g == 0), min(GRAPH_TICKS, steps))


class Clock:
    """R repetitions of exactly K steps, each bracketed by barrier + synchronize on both sides (the contract's timed
    region, repeated); HIP events on the launch stream give the device time of the same regions.  A rank's wall time runs
    from the opening barrier + synchronize to the synchronize that ends ITS K steps; the closing barrier follows the stamp
    and the MAX over ranks is what is reported (with the stamp behind the barrier every rank would report the same time plus
    the barrier's own latency, which is not part of the K steps and which N = 1 does not pay)."""

    def __init__(self, torch, dist, device=True):
        self.torch, self.dist, self.device = torch, dist, device   # device=False: the dry run (protocol only, no GPU)

    def fence(self):
        if self.device:
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.device:
            self.torch.cuda.synchronize()

    def measure(self, run_steps, steps, reps, min_total_s=0.03, max_reps=400):
        """Returns (wall seconds per repetition [max over ranks], device ms per repetition)."""
        torch = self.torch
        wall, dev = [], []

        def one():
            self.fence()
            if self.device:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            if self.device:
                e0.record()
            run_steps(steps)
            if self.device:
                e1.record()
                torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)   # this rank's K steps, complete on its device; the MAX over ranks is taken below
            self.fence()                            # closing barrier + synchronize: no rank starts the next region early
            dev.append(e0.elapsed_time(e1) if self.device else wall[-1] * 1e3)

        def over_ranks(values, op):
            if self.dist is None:
                return values
            tt = torch.tensor(values, dtype=torch.float64, device="cuda" if (self.device and self.dist.get_backend() == "nccl") else "cpu")
            self.dist.all_reduce(tt, op=op)
            return [float(v) for v in tt.cpu()]

        for _ in range(max(1, reps)):
            one()
        # short regions (a 20-step region of a launch-bound batch is 0.1 ms): more repetitions, the same number on every rank
        total = over_ranks([sum(wall)], self.dist.ReduceOp.MIN if self.dist is not None else None)[0]
        if total < min_total_s:
            more = min(max_reps - len(wall), int(math.ceil((min_total_s - total) / (total / len(wall)))))
            for _ in range(max(0, more)):
                one()
        wall = over_ranks(wall, self.dist.ReduceOp.MAX if self.dist is not None else None)
        return wall, dev


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def summarize(name, desc, models, dtype, batches, kernels, n_total, world, steps, wall, dev, launch_mode, extra_alg=0):
    """Common result record of a workload."""
    alg = sum(b.algorithmic_bytes * b.size for b in batches) + extra_alg
    elapsed = median(wall)
    dev_ms = median(dev)
    state = sum(b.resident_bytes_per_target * b.size for b in batches)
    return dict(name=name, desc=desc, model="+".join(models), dtype=dtype, targets_per_gpu=n_total,
                lanes_per_target=batches[0].lanes_per_target, layout="+".join(b.layout for b in batches),
                kernel="+".join(kernels), reps=len(wall), elapsed_s=elapsed, timed_total_s=sum(wall),
                ms_per_step=elapsed * 1e3 / steps, ms_per_step_min=min(wall) * 1e3 / steps, ms_per_step_max=max(wall) * 1e3 / steps,
                cycles_per_s=n_total * world * steps / elapsed, device_ms_per_step=dev_ms / steps,
                algorithmic_bytes_per_cycle=alg / n_total, algorithmic_bytes_per_step=alg,
                achieved_gbs=alg / (dev_ms * 1e-3 / steps) / 1e9, state_bytes=int(state),
                residency=("HBM-bound (state > 1 GB, 4 x the Infinity Cache)" if state > 4 * L3_BYTES else
                           "L3-assisted: state = %.1f x the 256 MB Infinity Cache (zig-zag traversal reuses the part touched last)" % (state / L3_BYTES) if state > L3_BYTES else
                           "L3-assisted: state fits the 256 MB Infinity Cache" if state > (32 << 20) else "L2/L3-resident, launch-bound"),
                launch_mode=launch_mode)


# ------------------------------------------------------------------------------------------------ workloads
def run_workload(te, torch, name, steps, warmup, lanes=0, targets=None, dist=None, rank=0, world=1, stream_ticks=64,
                 launch_mode="auto", reps=3, keep=False):
    """One motion model, one batch.  One launch of the step kernel per tick in every launch mode: "python" = one C-ABI
    call per tick; "sequence" = the launches of K ticks enqueued by one C call; "graph" = recorded hipGraph blocks
    replayed; "auto" = graph for launch-bound batches (<= SMALL targets), sequence otherwise; "fused" = K ticks in ONE
    launch (an "effective" figure, never `value`)."""
    import numpy as np
    from target_estimation_amd.streams import make_stream
    desc, model, dtype, n_targets, seed = WORKLOADS[name]
    if targets:
        n_targets = targets
    if not lanes:
        lanes = TUNED_LANES.get(name, 0)
    if launch_mode == "auto":
        launch_mode = "graph" if n_targets <= SMALL else "sequence"
    path = os.path.join(ROOT, "models", "model_%s_params.yaml" % model)
    mgr = te.TargetManager(path, dtype=dtype, lanes_per_target=lanes)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    mtype = te.MODEL_TYPES[model]
    dt = 1.0 / 250.0
    es = 8 if dtype == "f64" else 4
    # The synthetic measurements live in HBM as a ring of `ticks` ticks that the run cycles through.  Graph mode: every
    # gb-tick block of the ring is one recorded hipGraph.  Large batches: the ring is capped at RING_BYTES.
    gb = block_ticks(steps)
    if launch_mode in ("graph", "live"):
        ring_want = RINGS.get(name, stream_ticks)
        ticks = max(gb, min(ring_want, steps + warmup) // gb * gb)
    else:
        ticks = max(2, min(stream_ticks, steps + warmup, RING_BYTES // (7 * es * n_targets)))
    # ranks of a sharded run take consecutive target ranges of ONE keyed stream (csrc/stream_gen.hpp), in the batch precision
    st = make_stream(mtype, n_targets, ticks, dt, seed, dtype=dtype, first_target=rank * n_targets, **VARIANTS.get(name, {}))
    has = st["has_meas"]    # [ticks, N] uint8 or None
    ids = np.arange(n_targets, dtype=np.uint32) + rank * n_targets  # global ids: rank-contiguous shards
    if name in CLASSES:   # every target draws one of NC scaled copies of the model file's (Q, R, P0)
        nc = CLASSES[name]
        rng = np.random.default_rng(seed)
        prm = _model_params(model)
        sc = rng.uniform(0.5, 2.0, (nc, 3))
        mgr.init_batch_classes(ids, dt, 0.0, st["p0"].cpu().numpy(), mtype, prm["Q"][None] * sc[:, 0, None, None],
                               prm["R"][None] * sc[:, 1, None, None], prm["P"][None] * sc[:, 2, None, None],
                               (rng.integers(0, nc, n_targets) if name.endswith("_rand") else np.sort(rng.integers(0, nc, n_targets))).astype(np.uint32))
        assert len(mgr.batches()) == 1 and mgr.batches()[0].num_classes == nc
    else:
        mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy())
    b = mgr.batches()[0]
    meas = st["meas"]   # [ticks, 7, N], generated in the batch precision
    assert meas.dtype == b.torch_dtype()
    del st
    torch.cuda.synchronize()
    passes = 1
    if launch_mode == "graph" and ticks == gb:
        passes = next((q for q in range(GRAPH_PASSES, 0, -1) if steps % (gb * q) == 0), 1)
    done = [0]

    def run_ticks(count):
        if launch_mode == "python":
            for _ in range(count):
                b.step(dt, meas[done[0] % ticks], None if has is None else has[done[0] % ticks])
                done[0] += 1
            return
        if launch_mode == "sequence":   # `count` launches enqueued by one C call; tick s reads ring entry s % ticks
            b.step_sequence(dt, meas, has, use_graph=False, n_ticks=count)
            done[0] += count
            return
        while count > 0:
            off = done[0] % ticks
            if passes > 1 and off == 0 and count >= gb * passes:      # several passes over the ring in one graph
                b.step_sequence(dt, meas, has, use_graph=True, n_ticks=gb * passes)
                done[0] += gb * passes
                count -= gb * passes
                continue
            blk = min(count, gb - off % gb)
            if launch_mode == "fused":   # temporally fused: the whole block in ONE launch ("effective" metric)
                b.step_fused(dt, meas[off:off + blk], None if has is None else has[off:off + blk])
            else:   # only whole blocks are replayed from the recorded graphs; partial blocks go launch by launch
                b.step_sequence(dt, meas[off:off + blk], None if has is None else has[off:off + blk],
                                use_graph=(launch_mode == "graph" and off % gb == 0 and blk == gb))
            done[0] += blk
            count -= blk

    if launch_mode == "live":
        return run_live(te, torch, name, desc, model, dtype, mgr, b, meas, has, ids, dt, n_targets, world, steps, warmup, reps, ticks)
    if launch_mode == "graph":
        for off in range(0, ticks - gb + 1, gb):   # record every block's graph now (set-up; launches nothing)
            b.step_sequence(dt, meas[off:off + gb], None if has is None else has[off:off + gb], use_graph=2)
        if passes > 1:
            b.step_sequence(dt, meas, has, use_graph=2, n_ticks=gb * passes)
    run_ticks(warmup)
    done[0] = 0          # the timed region starts on a block boundary of the ring
    clock = Clock(torch, dist)
    wall, dev = clock.measure(run_ticks, steps, reps)
    x, P = mgr.get_state_batch(ids[:64])
    assert np.isfinite(x).all() and np.isfinite(P).all()
    res = summarize(name, desc, [model], dtype, [b], [kernel_name(b, model)], n_targets, world, steps, wall, dev, launch_mode)
    res.update(measurement_ring_ticks=ticks, measurement_ring_bytes=int(meas.numel() * meas.element_size()),
               survey_full_P_bytes_per_cycle=SURVEY_WORDS[model] * es)
    res["device_ms_per_launch"] = res["device_ms_per_step"]
    res["kernels"] = [dict(kernel=kernel_name(b, model), model=model, units_per_launch=n_targets,
                           algorithmic_bytes_per_unit=b.algorithmic_bytes, avg_launch_ms=res["device_ms_per_step"],
                           achieved_gbs=res["achieved_gbs"], frac=res["achieved_gbs"] / HBM_PEAK_GBS)]
    if keep:
        res["_mgr"] = (mgr, b, meas, ids, dt)
    else:
        mgr.close()
    return res


def run_live(te, torch, name, desc, model, dtype, mgr, b, meas, has, ids, dt, n_targets, world, steps, warmup, reps, ring_ticks):
    """Resident ("live") mode (target_batch_live_*): ONE launch holds the batch's state in registers; the host posts one tick per
    doorbell.  Two figures, both with exactly one tick per doorbell, neither is ever `value`:
      back_to_back  the host posts the K doorbells without waiting (as the per-tick launches of the other modes are enqueued
                    without waiting) and the region ends when every wavefront has finished tick K
      paced         the next doorbell is posted only after the previous tick's completion was seen on the host (a full
                    host <-> device round trip per tick: the latency of a stream with one tick in flight)
    The ring was filled in advance (it cycles); wall clock only -- there is no launch boundary to put device events on, and a
    device-wide synchronise would wait for the session."""
    import numpy as np
    torch.cuda.synchronize()
    b.live_start(dt, meas, has, max_ticks=1 << 30, idle_limit_s=10.0)   # (the resident kernel runs on a stream of its own)
    posted = [0]

    def back_to_back(count):
        b.live_post_each(count)
        posted[0] += count
        if not b.live_wait(posted[0], 20.0):
            raise RuntimeError("live session stalled at %d of %d" % (b.live_done(), posted[0]))

    def paced(count):
        for _ in range(count):
            b.live_post(1)
            posted[0] += 1
            if not b.live_wait(posted[0], 20.0):
                raise RuntimeError("live session stalled")

    def timed(fn, k, r):
        out = []
        for _ in range(r):
            t0 = time.perf_counter()
            fn(k)
            out.append(time.perf_counter() - t0)
        return out
    back_to_back(max(warmup, 64))
    k = max(steps, 4096)                  # a region of a few milliseconds
    wall = timed(back_to_back, k, max(reps, 5))
    wall_paced = timed(paced, min(k, 2000), 3)
    served = b.live_stop()
    assert served == posted[0]
    x, P = mgr.get_state_batch(ids[:64])
    assert np.isfinite(x).all() and np.isfinite(P).all()
    res = summarize(name, desc, [model], dtype, [b], [kernel_name(b, model) + " (LIVE variant)"], n_targets, world, k, wall, [w * 1e3 for w in wall],
                    "live: one resident launch, one tick per doorbell, back to back")
    res.update(measurement_ring_ticks=ring_ticks, measurement_ring_bytes=int(meas.numel() * meas.element_size()),
               survey_full_P_bytes_per_cycle=SURVEY_WORDS[model] * (8 if dtype == "f64" else 4),
               live=dict(ticks_per_region=k, us_per_tick_back_to_back=median(wall) / k * 1e6, us_per_tick_back_to_back_min=min(wall) / k * 1e6,
                         us_per_tick_paced=median(wall_paced) / min(k, 2000) * 1e6, ticks_served=served,
                         note="algorithmic HBM bytes per tick in this mode are the measurements only (the state never leaves the registers): "
                              "achieved_gbs / frac below are computed on the per-tick-launch byte count for comparison and are NOT a roofline claim"))
    res["device_ms_per_launch"] = res["device_ms_per_step"]
    res["kernels"] = [dict(kernel=kernel_name(b, model) + " (LIVE variant)", model=model, units_per_launch=n_targets,
                           algorithmic_bytes_per_unit=b.algorithmic_bytes, avg_launch_ms=res["device_ms_per_step"],
                           achieved_gbs=res["achieved_gbs"], frac=res["achieved_gbs"] / HBM_PEAK_GBS)]
    mgr.close()
    return res


def run_live_mixed(te, torch, name, desc, parts, dtype, mgr, batches, meas, dt, n_all, world, steps, warmup, reps, ring_ticks, query=None, outs=None):
    """run_live for every batch of a manager at once (target_manager_live_*_all): one resident kernel per motion model; with
    `query` the own-time sphere query of every target runs after every tick inside them (configs[4])."""
    torch.cuda.synchronize()
    mgr.live_start_all(dt, meas, max_ticks=1 << 30, idle_limit_s=10.0, query=query)
    posted = [0]

    def back_to_back(count):
        mgr.live_post_all(count, one_doorbell_per_tick=True)
        posted[0] += count
        if not mgr.live_wait_all(posted[0], 20.0):
            raise RuntimeError("live sessions stalled at %d of %d" % (mgr.live_done_all(), posted[0]))

    def paced(count):
        for _ in range(count):
            mgr.live_post_all(1)
            posted[0] += 1
            if not mgr.live_wait_all(posted[0], 20.0):
                raise RuntimeError("live sessions stalled")

    def timed(fn, k, r):
        out = []
        for _ in range(r):
            t0 = time.perf_counter()
            fn(k)
            out.append(time.perf_counter() - t0)
        return out
    back_to_back(max(warmup, 64))
    k = max(steps, 4096)
    wall = timed(back_to_back, k, max(reps, 5))
    wall_paced = timed(paced, min(k, 2000), 3)
    served = mgr.live_stop_all()
    assert served == posted[0]
    for b in batches:
        p, _, _ = b.get_est(twist=False, acc=False)
        assert torch.isfinite(p).all()
    models = [m for m, _ in parts]
    kernels = [kernel_name(b, m) + " (LIVE variant)" + ("+query" if query else "") for b, m in zip(batches, models)]
    res = summarize(name, desc, models, dtype, batches, kernels, n_all, world, k, wall, [w * 1e3 for w in wall],
                    "live: one resident launch per motion model, one tick per doorbell, back to back")
    if query:
        res["intersections_last_tick"] = sum(int((o[0] > -1).sum()) for o in outs)
    res.update(measurement_ring_ticks=ring_ticks, measurement_ring_bytes=int(sum(m.numel() * m.element_size() for m in meas)),
               live=dict(ticks_per_region=k, us_per_tick_back_to_back=median(wall) / k * 1e6, us_per_tick_back_to_back_min=min(wall) / k * 1e6,
                         us_per_tick_paced=median(wall_paced) / min(k, 2000) * 1e6, ticks_served=served,
                         note="the state never leaves the registers: achieved_gbs / frac are on the per-tick-launch byte count, for comparison only"))
    res["kernels"] = [dict(kernel=kn, model=m, units_per_launch=b.size, algorithmic_bytes_per_unit=b.algorithmic_bytes,
                           avg_launch_ms=res["device_ms_per_step"], achieved_gbs=res["achieved_gbs"], frac=res["achieved_gbs"] / HBM_PEAK_GBS)
                      for kn, m, b in zip(kernels, models, batches)]
    mgr.close()
    return res


def run_mixed(te, torch, name, steps, warmup, dist=None, rank=0, world=1, stream_ticks=64, launch_mode="auto", reps=3, split=1):
    """Several motion models (batches) in one manager, one step launch per batch per tick; the sphere query of
    configs[4] runs inside the step kernels.  Launch-bound populations: the batches are concurrent branches of ONE
    recorded hipGraph (target_manager_step_sequence_all).  Large populations: plain launches in tick order on the
    manager's stream (AR tick s, AV tick s, AR tick s+1, ...) -- the kernels are tens of microseconds long, and a
    serial chain is what rocprofv3's per-kernel durations can be compared with."""
    import numpy as np
    from target_estimation_amd.streams import make_stream
    desc, parts, dtype, seed, intersect = MIXED[name]
    if split > 1:   # strong scaling: this rank's share of every model (N_model / N per GPU per model)
        parts = [(m, n // split) for m, n in parts]
    n_all = sum(n for _, n in parts)
    if launch_mode == "auto":
        launch_mode = "graph" if n_all <= SMALL else "sequence"
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = 1.0 / 250.0
    es = 8 if dtype == "f64" else 4
    gb = block_ticks(steps)
    if launch_mode in ("graph", "live"):
        ticks = gb
    else:
        ticks = max(2, min(stream_ticks, steps + warmup, RING_BYTES // (7 * es * n_all)))
    streams, base = [], 0
    for k, (model, n) in enumerate(parts):
        mt = te.MODEL_TYPES[model]
        st = make_stream(mt, n, ticks, dt, seed + 17 * k, dtype=dtype, first_target=rank * n)
        ids = np.arange(n, dtype=np.uint32) + base + rank * 16_000_000
        base += n
        params = _model_params(model)
        mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy(), None, None, type=mt, Q=params["Q"], R=params["R"], P0=params["P"])
        streams.append(st["meas"])
        del st
    batches = mgr.batches()
    assert len(batches) == len(parts)
    meas = streams
    assert all(m.dtype == b.torch_dtype() for m, b in zip(meas, batches))
    origin = np.zeros(3)
    outs = [(torch.empty(b.size, dtype=torch.float64, device="cuda"), torch.empty((b.size, 7), dtype=torch.float64, device="cuda"))
            for b in batches] if intersect else None
    query = (origin, 1.0, [o[0] for o in outs], [o[1] for o in outs]) if intersect else None
    passes = next((q for q in range(GRAPH_PASSES, 0, -1) if steps % (gb * q) == 0), 1) if launch_mode == "graph" else 1

    def run(count):
        if launch_mode == "python":
            for s in range(count):
                for j, b in enumerate(batches):
                    b.step(dt, meas[j][s % ticks])
                    if intersect:
                        mgr._lib.target_batch_intersect_sphere_dev(b._h, float("nan"), origin.ctypes.data_as(te.capi.c_double_p), 1.0,
                                                                   outs[j][0].data_ptr(), outs[j][1].data_ptr())
            return
        if launch_mode == "sequence":
            mgr.step_sequence_all(dt, meas, query=query, use_graph=0, n_ticks=count)
            return
        while count > 0:
            if passes > 1 and count >= gb * passes:
                mgr.step_sequence_all(dt, meas, query=query, use_graph=1, n_ticks=gb * passes)
                count -= gb * passes
            elif count >= gb:
                mgr.step_sequence_all(dt, meas, query=query, use_graph=1)
                count -= gb
            else:
                mgr.step_sequence_all(dt, [m[:count] for m in meas], query=query, use_graph=0)
                count = 0

    if launch_mode == "live":
        return run_live_mixed(te, torch, name, desc, parts, dtype, mgr, batches, meas, dt, n_all, world, steps, warmup, reps, ticks, query, outs)
    if launch_mode == "graph":
        mgr.step_sequence_all(dt, meas, query=query, use_graph=2)    # record before the timed region
        if passes > 1:
            mgr.step_sequence_all(dt, meas, query=query, use_graph=2, n_ticks=gb * passes)
    run(warmup)
    clock = Clock(torch, dist)
    wall, dev = clock.measure(run, steps, reps)
    models = [m for m, _ in parts]
    kernels = [kernel_name(b, m) + ("+query" if intersect else "") for b, m in zip(batches, models)]
    extra_alg = sum(8 * 8 * b.size for b in batches) if intersect else 0   # the query writes delta + pose7 (doubles)
    res = summarize(name, desc, models, dtype, batches, kernels, n_all, world, steps, wall, dev, launch_mode, extra_alg)
    if launch_mode in ("graph", "sequence") and mgr.population_tick():
        # ONE launch steps every batch of the manager (csrc/kf_step_sep.hpp kf_step_population_kernel): the launch's duration is
        # the tick's device time over the timed region itself (HIP events on the launch stream), its algorithmic bytes the sum
        # over the models of targets x bytes per target -- no attribution pass, nothing to split
        kname = "kf_step_population_kernel<%s>" % ("double" if dtype == "f64" else "float") + ("+query" if intersect else "")
        per_unit = res["algorithmic_bytes_per_step"] / n_all
        ms = res["device_ms_per_step"]
        res["kernel"] = kname
        res["launch_mode"] = launch_mode + ": one launch per tick for the whole population"
        res["kernels"] = [dict(kernel=kname, model="+".join(models), units_per_launch=n_all, algorithmic_bytes_per_unit=per_unit,
                               avg_launch_ms=ms, achieved_gbs=per_unit * n_all / (ms * 1e-3) / 1e9,
                               frac=per_unit * n_all / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               parts=[dict(model=m, units=b.size, algorithmic_bytes_per_unit=b.algorithmic_bytes + (64 if intersect else 0))
                                      for b, m in zip(batches, models)],
                               note="one launch = one tick of every batch; algorithmic_bytes_per_unit is the mean over the population "
                                    "(sum over parts of units x bytes / units_per_launch)" + ("; the query's delta + pose7 outputs (64 B per target) included" if intersect else ""))]
        for b in batches:
            p, _, _ = b.get_est(twist=False, acc=False)
            assert torch.isfinite(p).all()
        if intersect:
            res["intersections_last_tick"] = sum(int((o[0] > -1).sum()) for o in outs)
        res["measurement_ring_ticks"] = ticks
        res["measurement_ring_bytes"] = int(sum(m.numel() * m.element_size() for m in meas))
        mgr.close()
        return res
    # Attribution pass (same kernels, same data, same stream, same launch order as the timed region, right after it): one
    # HIP event between consecutive launches -> average duration of every kernel IN ITS CONTEXT (alone, a 500 000-target
    # batch would sit in the Infinity Cache and look faster than it is inside the tick).
    k_steps = max(2, min(steps, 32) // 2 * 2)
    nb = len(batches)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(k_steps * nb + 1)]
    slots = []
    torch.cuda.synchronize()
    evs[0].record()
    for s in range(k_steps):
        for j in (range(nb) if s % 2 == 0 else range(nb - 1, -1, -1)):   # zig-zag over the whole tick, as stepSequenceAll does
            batches[j].step(dt, meas[j][s % ticks])
            slots.append(j)
            evs[len(slots)].record()
    torch.cuda.synchronize()
    per = [[] for _ in range(nb)]
    for i, j in enumerate(slots):
        per[j].append(evs[i].elapsed_time(evs[i + 1]))
    # An event between two launches costs a few microseconds of its own, so the periods above overstate the kernels.  What
    # is exact is the tick's device time over the timed region; it is split between the kernels in the ratio of their
    # event periods (the launch boundaries inside the tick stay included).
    period = [sum(p) / len(p) for p in per]
    res["kernels"] = []
    for j, (b, m) in enumerate(zip(batches, models)):
        ms = res["device_ms_per_step"] * period[j] / sum(period)
        per_unit = b.algorithmic_bytes
        res["kernels"].append(dict(kernel=kernel_name(b, m), model=m, units_per_launch=b.size, algorithmic_bytes_per_unit=per_unit,
                                   avg_launch_ms=ms, achieved_gbs=per_unit * b.size / (ms * 1e-3) / 1e9,
                                   frac=per_unit * b.size / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   event_period_ms=period[j],
                                   note=("share measured without the fused query" if intersect else "")))
    for b in batches:
        p, _, _ = b.get_est(twist=False, acc=False)
        assert torch.isfinite(p).all()
    if intersect:
        res["intersections_last_tick"] = sum(int((o[0] > -1).sum()) for o in outs)
    res["measurement_ring_ticks"] = ticks
    res["measurement_ring_bytes"] = int(sum(m.numel() * m.element_size() for m in meas))
    mgr.close()
    return res


# ------------------------------------------------------------------------------------------------ CPU baseline, parity
def host_threads(omp_max):
    """Threads the CPU baseline may use: the smallest of OpenMP's default, the affinity mask and
    the cgroup CPU quota (a GPU box hands one GPU a 16-CPU share of a larger host)."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("TE_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_time_model(oracle, model, dtype, n_cpu, threads, budget_s, seed):
    """cycles/s of the oracle port on `n_cpu` targets of one model with `threads` OpenMP threads."""
    import numpy as np
    m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_%s_params.yaml" % model))
    rng = np.random.default_rng(seed)
    p0 = np.concatenate([rng.uniform(-10, 10, (n_cpu, 3)), np.tile([0, 0, 0, 1.0], (n_cpu, 1))], 1)
    dt = 1.0 / 250.0
    ob = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype, fast=True)
    meas = p0.copy()
    ob.step(dt, meas, nthreads=threads)  # warm
    t0 = time.perf_counter()
    ticks = 0
    while True:
        meas[:, :3] += 0.004 + rng.normal(0, 0.01, (n_cpu, 3))
        ob.step(dt, meas, nthreads=threads)
        ticks += 1
        if time.perf_counter() - t0 > budget_s and ticks >= 3:
            break
    return n_cpu * ticks / (time.perf_counter() - t0), ticks


def cpu_baseline(parts, dtype, budget_s=5.0, per_model=True):
    """The CPU oracle (oracle/, the 'port' of the reference's Eigen path; the reference itself cannot be built here: no
    Eigen3) with OpenMP static over targets on this box's host cores, rebuilt here with -march=native, on a BOUNDED
    sample of the same workload: the same model mix, 10 000 targets per model."""
    import oracle
    try:   # build for THIS host's cores (the shipped .so was built -march=native in the build container)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "_build/libte_oracle_fast.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rebuilt = True
    except Exception:
        rebuilt = False
    threads = host_threads(oracle.load(True).orc_max_threads())
    n_cpu = 10_000
    models = [m for m, _ in parts]
    weights = [n for _, n in parts]
    rates, samples = {}, []
    for k, model in enumerate(models):
        r, ticks = _cpu_time_model(oracle, model, dtype, n_cpu, threads, budget_s / len(models), 100 + k)
        rates[model] = r
        samples.append("%d %s x %d ticks" % (n_cpu, model, ticks))
    # cycles/s of the mix = total cycles / total time, time per model weighted by its share of the population
    tot = sum(weights)
    value = tot / sum(w / rates[m] for m, w in zip(models, weights))
    r1, _ = _cpu_time_model(oracle, models[0], dtype, 2000, 1, 1.0, 7)
    out = dict(value=value, unit="cycles/s", cores=int(threads), kind="port",
               sample="%s (%s, OpenMP static over targets, -O3 -march=native %s)" % (" + ".join(samples), dtype,
                                                                                      "rebuilt on this host" if rebuilt else "as shipped"),
               cpu_model=cpu_model(), value_1thread_first_model=r1)
    if per_model:   # one figure per YAML motion model in the reference's precision, beside the per-model extras
        pm = {}
        for model in MODEL_N:
            if model in rates and dtype == "f64":
                pm[model] = rates[model]
            else:
                pm[model], _ = _cpu_time_model(oracle, model, "f64", n_cpu, threads, 1.2, 11)
        out["per_model_f64"] = pm
    return out


def parity_report(te, torch, parts, dtype, seed, n_sample=256, checkpoints=(1, 100, 1000)):
    """parity next to the perf number.  A sample of the same workload (same models, precision, layout choice
    and stream generator) stepped on the GPU and by the CPU oracle; errors after 1 / 100 / 1000 ticks.  The oracle is
    the checker here, nothing of it is timed."""
    import numpy as np
    import oracle
    from target_estimation_amd.streams import make_stream
    rep = dict(targets_per_model=n_sample, ticks=list(checkpoints), oracle="oracle/te_oracle.c, same precision (%s); parity UNPINNED: "
               "the reference cannot be built here (no Eigen3), the oracle is a restatement; both sides generate the keyed stream "
               "themselves (target_stream_fill_dev on the GPU, oracle/te_stream.c on the CPU)" % dtype,
               tolerance=("x: 1e-10 + 1e-10|x|, P: 1e-9 max|P|" if dtype == "f64" else "x: 2e-3 + 1e-4|x|, P: 2e-3 max|P|"), models={})
    dt, ticks = 1.0 / 250.0, max(checkpoints)
    for k, (model, _) in enumerate(parts):
        m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_%s_params.yaml" % model))
        st = make_stream(te.MODEL_TYPES[model], n_sample, ticks, dt, seed + 17 * k, dtype=dtype)
        ref = oracle.stream_fill(te.MODEL_TYPES[model], seed + 17 * k, n_sample, ticks, dt, dtype=dtype)   # the CPU regenerates the stream
        p0 = ref["p0"]
        ids = np.arange(n_sample, dtype=np.uint32)
        mgr = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % model), dtype=dtype)
        mgr.set_stream(torch.cuda.current_stream().cuda_stream)
        mgr.init_batch(ids, dt, 0.0, p0)
        b = mgr.batches()[0]
        meas = st["meas"]
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype)
        r = dict(layout=b.layout, ids_exact=bool((b.slot_ids() == ids).all()), max_abs_x=[], max_rel_P=[])
        for s in range(ticks):
            b.step(dt, meas[s])
            orc.step(dt, ref["meas"][s])
            if s + 1 in checkpoints:
                x, P = mgr.get_state_batch(ids)
                xo, Po = orc.state()
                r["max_abs_x"].append(float(np.abs(x - xo).max()))
                r["max_rel_P"].append(float((np.abs(P - Po) / np.abs(Po).max(axis=(1, 2), keepdims=True)).max()))
        rep["models"][model] = r
        mgr.close()
    return rep


class GatherTimeout(TimeoutError):
    pass


def configs0_report(te, torch, steps=(1000, 10000)):
    """BASELINE.json configs[0]: 1 target, uniform-velocity model file, the reference integration test's loop (init, then per
    step update(id, dt, meas) + getTargetPose + getTargetTwist; test/target_manager_test.cpp:125-146) on the reference
    test's own stream.  CPU: the oracle port, one thread, the loop in C (oracle.harness_run).  GPU: the same loop through
    the library's ten reference symbols (one target: a latency path, reported for completeness, never `value`)."""
    import ctypes as C
    import numpy as np
    import oracle
    m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_uniform_velocity_params.yaml"))
    dt = 1.0 / m["frequency"]
    stream = oracle.ref_test_stream(n_models=1)[0]           # [10000, 7]
    out = dict(workload="1 target, model_uniform_velocity_params.yaml, the reference test's stream (libstdc++ default_random_engine, "
               "N(0, 0.01), line to (0.2, 0.3, 0.4)); per step: update + getTargetPose + getTargetTwist", dt=dt, rows=[])
    lib = te.capi.lib()
    path = os.path.join(ROOT, "models", "model_uniform_velocity_params.yaml").encode()
    for n in steps:
        meas = np.ascontiguousarray(stream[:n])
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            pose_c, twist_c = oracle.harness_run(m["model"], m["Q"], m["R"], m["P"], meas, dt)
            best = min(best, time.perf_counter() - t0)
        h = lib.target_manager_new(path)
        p0 = np.ascontiguousarray(meas[0])
        lib.target_manager_init(h, 0, dt, p0.ctypes.data_as(te.capi.c_double_p), 0.0)
        pose = np.zeros((n, 7)); twist = np.zeros((n, 6))
        mp, pp, tp = (a.ctypes.data_as(te.capi.c_double_p) for a in (meas, pose, twist))
        dbl = C.sizeof(C.c_double)
        t0 = time.perf_counter()
        for i in range(n):
            lib.target_manager_update_meas(h, 0, dt, C.cast(C.addressof(mp.contents) + 7 * dbl * i, te.capi.c_double_p))
            lib.target_manager_get_est_pose(h, 0, C.cast(C.addressof(pp.contents) + 7 * dbl * i, te.capi.c_double_p))
            lib.target_manager_get_est_twist(h, 0, C.cast(C.addressof(tp.contents) + 6 * dbl * i, te.capi.c_double_p))
        gpu_s = time.perf_counter() - t0
        lib.target_manager_delete(h)
        out["rows"].append(dict(steps=n, cpu_oracle_1thread_s=best, cpu_cycles_per_s=n / best, cpu_us_per_step=best / n * 1e6,
                                gpu_ten_symbol_abi_s=gpu_s, gpu_us_per_step=gpu_s / n * 1e6, gpu_cycles_per_s=n / gpu_s,
                                end_position_gpu=pose[-1, :3].tolist(), end_position_cpu=pose_c[-1, :3].tolist(),
                                max_abs_pose_difference=float(np.abs(pose - pose_c).max()),
                                max_abs_twist_difference=float(np.abs(twist - twist_c).max())))
    return out


def start_watchdog(what, rank, deadline_s, grace_s=5.0):
    """A daemon thread that ends THIS process (status 3, with a diagnostic) unless the returned event is set within
    deadline_s + grace_s: whatever `what` is -- a collective entry a peer never reaches, a send that never completes -- a
    rank that is alone in it must neither hang the job nor be re-executed.  Returns (event, absolute deadline)."""
    done = threading.Event()
    t_end = time.monotonic() + deadline_s

    def watchdog():
        if not done.wait(deadline_s + grace_s):
            print("bench.py rank %d: %s still not finished %.0f s after its start (communicator set-up or a peer's "
                  "send/recv never completed); giving up" % (rank, what, deadline_s + grace_s), file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    return done, t_end


def gather_rehearsal(dist, rank, world, deadline_s):
    """--dry-run with N > 1: the gather's collective entry (the communicator set-up of gather_report is one) under the same
    watchdog, with a barrier standing for it.  TE_BENCH_TEST_LATE_RANK=R:SECONDS makes rank R arrive that late (tests)."""
    late = os.environ.get("TE_BENCH_TEST_LATE_RANK", "")
    if late:
        r, sec = late.split(":")
        if int(r) == rank:
            time.sleep(float(sec))   # still busy with something else: it has not entered the gather yet
    done, _ = start_watchdog("pose gather", rank, deadline_s, grace_s=1.0)
    dist.barrier()
    done.set()


def gather_report(te, torch, dist, rank, world, workload="ar1m64", ticks=16, deadline_s=60.0):
    """The library's RCCL pose gather (target_manager_gather_pose_*: direct sends to rank 0 on a second stream behind an
    event).  Exposed = begin + wait with nothing else running; overlapped = begin, then `ticks` ticks, then wait: what
    the ticks cost on top of their own time is what the gather did NOT hide."""
    from target_estimation_amd.dist import PoseGather
    r = run_workload(te, torch, workload, ticks, 4, dist=dist, rank=rank, world=world, reps=2, keep=True)
    mgr, b, meas, ids, dt = r.pop("_mgr")
    # Bounded: a peer that never arrives (communicator set-up is collective) must not hang the run.  A watchdog thread ends
    # the process with a diagnostic after deadline_s; the waits themselves poll hipEventQuery against the same deadline
    # (target_manager_gather_pose_wait_for).  Nothing is re-executed.
    done, t_end = start_watchdog("pose gather", rank, deadline_s)
    g = PoseGather(mgr)
    g.deadline = t_end
    counts = g.counts()
    clock = Clock(torch, dist)

    def timed(fn):
        clock.fence()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def only_gather():
        g.begin(counts)
        g.wait()

    def only_ticks():
        b.step_sequence(dt, meas, None, use_graph=False, n_ticks=ticks)

    def both():
        g.begin(counts)
        b.step_sequence(dt, meas, None, use_graph=False, n_ticks=ticks)
        g.wait()

    only_gather(); only_ticks()
    exposed = min(timed(only_gather) for _ in range(5))
    t_ticks = min(timed(only_ticks) for _ in range(5))
    t_both = min(timed(both) for _ in range(5))
    done.set()
    out = dict(name="gather_pose", workload=workload, n_gpus=world, comm_ranks=g.world, rows_per_rank=counts[0], rows_total=sum(counts),
               bytes_to_root=sum(counts) * 56, gather_pose_ms_exposed=exposed, ticks=ticks, ticks_ms=t_ticks,
               ticks_plus_gather_ms=t_both, gather_pose_ms_overlapped=max(0.0, t_both - t_ticks),
               transport="RCCL ncclSend/ncclRecv to rank 0 on a second stream" if world > 1 else "world size 1: the root's own rows only (no peer traffic)")
    g.close()
    mgr.close()
    return out


def copy_bandwidth(torch, nbytes=1 << 30, reps=10):
    """Streaming rates of this box in GB/s (bytes read + bytes written per second), the practical HBM ceilings SURVEY 8d
    asks to record next to the 8 TB/s spec peak: a device-to-device copy (dst.copy_(src): 1 read + 1 write) and a triad
    (c = a + b on fp32: 2 reads + 1 write), both over 1 GiB arrays (larger than the 256 MB Infinity Cache)."""
    n = nbytes // 4
    a = torch.zeros(n, dtype=torch.float32, device="cuda")
    b = torch.ones(n, dtype=torch.float32, device="cuda")
    c = torch.empty_like(a)

    def rate(fn, bytes_per_call):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return bytes_per_call * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return dict(copy=rate(lambda: c.copy_(a), 2.0 * nbytes), triad=rate(lambda: torch.add(a, b, out=c), 3.0 * nbytes))


def previous_side_file():
    """The newest committed side file of an EARLIER collection (profiles/rNN_bench_extra*.json), for the slow-down check below."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_extra*.json")):
        m = re.match(r"r(\d\d)_bench_extra(_head)?\.json$", os.path.basename(f))
        if m:
            key = (int(m.group(1)), 1 if m.group(2) else 0)
            if best is None or key > best[0]:
                best = (key, f)
    if best is None:
        return None, {}
    try:
        d = json.load(open(best[1]))
    except (OSError, ValueError):
        return None, {}
    rows = {e["name"]: e["ms_per_step"] for e in d.get("extra", []) if "ms_per_step" in e}
    if "line" in d:
        rows[d["line"]["config"]["name"]] = d["line"]["ms_per_step"]
    return os.path.relpath(best[1], ROOT), rows


def slowdown_check(rows_now, factor=1.25):
    """Workloads that take more than `factor` x their time in the committed previous side file: one stderr line each, and the
    list for the side file.  (Round 3 shipped a 2-3x slow-down of three rows that no record said anything about.)"""
    src, before = previous_side_file()
    slow = []
    for name, ms in rows_now.items():
        if name in before and before[name] > 0 and ms > factor * before[name]:
            slow.append(dict(name=name, ms_per_step=ms, previous_ms_per_step=before[name], ratio=ms / before[name], previous_record=src))
            print("bench.py: WARNING: %s takes %.4f ms per step, %.2f x its %.4f ms in %s" % (name, ms, ms / before[name], before[name], src),
                  file=sys.stderr, flush=True)
    return slow


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json, tools/pmc_traffic.py):
    {workload: {kernel: {hbm_read_bytes, hbm_write_bytes}}}.  PMC counters cannot be read from inside this process."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except (OSError, ValueError):
        return {}


def attach_traffic(kernels, workload, traffic):
    rows = traffic.get(workload, {}) if not workload.startswith("_") else {}
    for k in kernels:
        t = rows.get(k["kernel"])
        k["traffic"] = (t["hbm_read_bytes"] + t["hbm_write_bytes"]) if t else None


def make_line(args, res, dom, world, traffic):
    # THE LINE: what the contract asks for and nothing else (the driver parses the last stdout line out of a bounded tail;
    # round 2's 30 KB line did not parse).  Everything else goes to the side file named in config.side_file and to stderr.
    out = {
        "metric": "KF predict+update cycles/sec over N targets",
        "value": res["cycles_per_s"], "unit": "cycles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic",
        "config": {"workload": res["desc"], "name": res["name"], "motion_model": res["model"],
                   "targets_per_gpu": res["targets_per_gpu"], "targets_total": res["targets_per_gpu"] * world,
                   "P_layout": res["layout"], "launch_mode": res["launch_mode"],
                   "state_bytes_per_gpu": res["state_bytes"], "residency": res["residency"].split(":")[0].split(" (")[0],
                   "side_file": os.path.relpath(side_path(args), ROOT)},
        "roofline": {"bound": "hbm", "achieved": dom["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["achieved_gbs"] / HBM_PEAK_GBS, "traffic": dom.get("traffic"),
                     "kernel": dom["kernel"], "algorithmic_bytes_per_unit": dom["algorithmic_bytes_per_unit"],
                     "units_per_launch": dom["units_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
                     "tick_frac": res["achieved_gbs"] / HBM_PEAK_GBS,
                     "traffic_source": traffic_source(traffic) if dom.get("traffic") else None},
    }
    return out


LINE_LIMIT = 4096   # bytes of the one stdout line (the driver keeps a bounded tail of stdout and parses its last line)


def side_path(args):
    return os.path.abspath(args.side_file or os.environ.get("TE_BENCH_SIDE") or os.path.join(ROOT, "bench_extra.json"))


def write_side(args, obj):
    """Everything that is not the line (extras, per-kernel table, parity sample, configs[0], gather): one JSON file."""
    path = side_path(args)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path + ".tmp", "w") as f:
            json.dump(obj, f, indent=1)
        os.replace(path + ".tmp", path)
    except OSError as exc:
        print("bench.py: could not write %s: %s" % (path, exc), file=sys.stderr)


def compact_line(out):
    """The line as it will be printed: floats to 6 significant digits, and never longer than LINE_LIMIT -- optional keys are
    dropped (last resort) rather than printing a line the driver cannot parse."""
    def rnd(o):
        if isinstance(o, float):
            return float("%.6g" % o) if math.isfinite(o) else None
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [rnd(v) for v in o]
        return o
    line = rnd(out)
    for drop in (None, "parity", ("config", "residency"), ("roofline", "traffic_source"), ("cpu_baseline", "sample")):
        if drop is not None:
            if isinstance(drop, tuple):
                line.get(drop[0], {}).pop(drop[1], None)
            else:
                line.pop(drop, None)
        if len(json.dumps(line)) + 1 <= LINE_LIMIT:
            break
    assert len(json.dumps(line)) + 1 <= LINE_LIMIT, "bench line too long"
    return line


def traffic_source(traffic):
    meta = traffic.get("_meta", {})
    return "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes) collected at commit %s" % meta.get("commit", "unknown")


def parity_summary(rep):
    """One short string for the line; the full sample is in the side file."""
    if "models" not in rep:
        return "partial (oracle unpinned)"
    dx = max(max(r["max_abs_x"]) for r in rep["models"].values())
    dP = max(max(r["max_rel_P"]) for r in rep["models"].values())
    ids = all(r["ids_exact"] for r in rep["models"].values())
    return "partial: oracle unpinned (reference unbuildable here); GPU vs oracle on %d targets/model after %s ticks: max|dx| %.2g, max|dP|/max|P| %.2g, ids %s" % (
        rep["targets_per_model"], "/".join(str(t) for t in rep["ticks"]), dx, dP, "exact" if ids else "MISMATCH")


_MAPS = os.environ.get("TE_BENCH_MAPS")   # diagnostic: keep /proc/self/maps current on disk (to resolve the PCs of a profiler-side abort)


def maps_checkpoint(tag):
    if not _MAPS:
        return
    try:
        with open("/proc/self/maps") as f, open(_MAPS + ".tmp", "w") as g:
            g.write("# after workload %s\n" % tag)
            g.write(f.read())
        os.replace(_MAPS + ".tmp", _MAPS)
        with open(_MAPS + ".progress", "a") as g:
            g.write("%s done\n" % tag)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------ rank start-up
def spawn_ranks(args, argv):
    """`--gpus N` without a torch.distributed environment: start the N ranks as a CHILD process group (this parent has
    not touched the GPU and never will) and pass its output and exit code through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TE_BENCH_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


def init_ranks(args, torch):
    """Process group of the run: returns (dist module or None, rank, ranks the group really has)."""
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_env != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world_env), file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("TE_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo rehearses the path on a one-GPU box
    if not args.dry_run:
        ndev = torch.cuda.device_count()
        if backend == "nccl" and world_env > max(1, ndev):
            if rank == 0:
                print("bench.py: --gpus %d but only %d GPU(s) are visible (one rank per GPU over RCCL)" % (args.gpus, ndev), file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank % max(1, ndev))
    if world_env == 1:
        return None, 0, 1
    import torch.distributed as dist
    dist.init_process_group("gloo" if args.dry_run else backend)
    ones = torch.ones(1, dtype=torch.int64)
    if dist.get_backend() == "nccl":
        ones = ones.cuda()
    dist.all_reduce(ones)   # every rank contributes 1: the sum is the number of ranks that really took part
    seen = int(ones.item())
    if seen != args.gpus or dist.get_world_size() != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, seen), file=sys.stderr)
        sys.exit(2)
    return dist, rank, seen


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region (median reported); more are taken "
                    "until they add up to 30 ms")
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS) + sorted(MIXED))
    ap.add_argument("--lanes", type=int, default=0, help="lanes per target / layout code (0 = the workload's default)")
    ap.add_argument("--targets", type=int, default=0, help="override targets per GPU (single-model workloads)")
    ap.add_argument("--extra", default=DEFAULT_EXTRA, help="comma list of extra workloads reported under 'extra' (N=1; '' = none)")
    ap.add_argument("--extra-multi", default=DEFAULT_EXTRA_MULTI,
                    help="extra workloads when --gpus > 1 (per-GPU sizes, every rank in lockstep; NAME_strong = the workload's targets split over the ranks)")
    ap.add_argument("--extra-steps", type=int, default=24)
    ap.add_argument("--stream-ticks", type=int, default=0, help="ticks of synthetic measurements kept in HBM and cycled through; 0 = default")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gather", action="store_true", help="(kept for compatibility: the gather report is on by default)")
    ap.add_argument("--no-gather", action="store_true", help="skip the gather report (it runs AFTER the line has been printed when N > 1)")
    ap.add_argument("--gather-deadline", type=float, default=60.0, help="seconds the pose gather may take before the run gives up (exit 3)")
    ap.add_argument("--post-deadline", type=float, default=1500.0, help="seconds the extras and the gather may take AFTER the line has been "
                    "printed before every rank gives up on them (exit 0)")
    ap.add_argument("--side-file", default="", help="where everything that is not the line goes (default bench_extra.json next to bench.py)")
    ap.add_argument("--launch-mode", default="auto", choices=["auto", "python", "sequence", "graph", "fused", "live"],
                    help="how the per-tick launches are enqueued (always one kernel launch per batch per tick, except 'fused')")
    ap.add_argument("--dry-run", action="store_true", help="no device work: rank start-up, rendezvous, barriers, timing protocol and "
                    "the JSON line only (value is null); what the CPU tests exercise")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: libraries that chat on stdout (RCCL prints a version banner when a
    # communicator is created) are sent to stderr for the whole run; the line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    dist, rank, world = init_ranks(args, torch)

    if args.dry_run:
        wall, _ = Clock(torch, dist, device=False).measure(lambda k: time.sleep(0.0005 * k), args.steps, args.reps)
        if dist is not None:
            dist.barrier()
        line = {"metric": "KF predict+update cycles/sec over N targets", "value": None, "unit": "cycles/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": median(wall) * 1e3 / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "dry-run: no device work", "config": {"workload": "dry-run of the rank start-up and timing protocol"}}
        if rank == 0:
            emit(line)   # first, as in a real run: nothing behind the line can cost it
        if dist is not None:
            if not args.no_gather:
                gather_rehearsal(dist, rank, world, args.gather_deadline)
            dist.barrier()
            dist.destroy_process_group()
        return

    import target_estimation_amd as te
    st_kw = {"stream_ticks": args.stream_ticks} if args.stream_ticks else {}
    if args.workload in MIXED:
        res = run_mixed(te, torch, args.workload, args.steps, args.warmup, dist, rank, world,
                        launch_mode="auto" if args.launch_mode in ("fused", "live") else args.launch_mode, reps=args.reps, **st_kw)
        parts, seed = MIXED[args.workload][1], MIXED[args.workload][3]
    else:
        res = run_workload(te, torch, args.workload, args.steps, args.warmup, args.lanes, args.targets or None,
                           dist, rank, world, launch_mode=args.launch_mode, reps=args.reps, **st_kw)
        parts, seed = [(WORKLOADS[args.workload][1], res["targets_per_gpu"])], WORKLOADS[args.workload][4]
    traffic = load_traffic()
    attach_traffic(res["kernels"], args.workload, traffic)
    dom = max(res["kernels"], key=lambda k: k["units_per_launch"] * k["algorithmic_bytes_per_unit"])
    out = make_line(args, res, dom, world, traffic)
    side = {
        "line_of": "bench.py " + " ".join(sys.argv[1:]),
        "config": {"dt": 0.004,
                   "timing": "median of %d repetitions of %d steps (min %.4f, max %.4f ms/step; %.1f ms timed in all)" % (
                       res["reps"], args.steps, res["ms_per_step_min"], res["ms_per_step_max"], res["timed_total_s"] * 1e3),
                   "measurements": "synthetic (counter-based generator of the library, csrc/stream_gen.hpp), resident in HBM as a ring of "
                                   "%d ticks (%d MB) that the run cycles through" % (
                                       res["measurement_ring_ticks"], res["measurement_ring_bytes"] // 1000000),
                   "residency": res["residency"], "sharding": "contiguous id ranges per rank, no data-path collective"},
        "roofline": {"kernel_note": "dominant kernel of the tick (largest share of the bytes).  avg_launch_ms = the tick's device time over the "
                                    "timed region (HIP events on the launch stream) x this kernel's share of it; the share comes from one HIP event "
                                    "between consecutive launches in a pass with the timed region's launch order, right after it",
                     "bytes_rule": "bytes the kernel reads + writes per target: 2n + 2|P stored| + measurement words read (3 linear, 7 angular) "
                                   "(+6 unwrap words, angular); SURVEY 8d's full-P figure for this model is survey_full_P_bytes_per_unit",
                     "survey_full_P_bytes_per_unit": (sum(q["units"] * SURVEY_WORDS[q["model"]] for q in dom["parts"]) / dom["units_per_launch"]
                                                      if "parts" in dom else SURVEY_WORDS[dom["model"]]) * (8 if res["dtype"] == "f64" else 4),
                     "tick": {"achieved": res["achieved_gbs"], "frac": res["achieved_gbs"] / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_step": res["algorithmic_bytes_per_step"], "device_ms_per_step": res["device_ms_per_step"],
                              "note": "all kernels of the tick over the timed region itself (events on the launch stream)"},
                     "kernels": res["kernels"]},
    }
    if world == 1 and rank == 0 and not args.no_cpu:
        try:
            cb = cpu_baseline(parts, res["dtype"])
            out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
            side["cpu_baseline"] = cb
        except Exception as exc:
            out["cpu_baseline"] = {"error": str(exc)[:200]}
        try:
            side["parity"] = parity_report(te, torch, parts, res["dtype"], seed)
            out["parity"] = parity_summary(side["parity"])
        except Exception as exc:   # a diagnostic: never let it break the bench line
            side["parity"] = {"error": str(exc)[:300]}
        try:
            side["configs0"] = configs0_report(te, torch)
        except Exception as exc:
            side["configs0"] = {"error": str(exc)[:300]}
    if world == 1 and rank == 0:
        try:
            bw = copy_bandwidth(torch)
            side["roofline"]["measured_copy_gbs"] = bw["copy"]
            side["roofline"]["measured_triad_gbs"] = bw["triad"]
            # fractions of the box's own streaming rates next to the fraction of the 8 TB/s nominal peak
            side["roofline"]["tick"]["frac_of_measured_copy"] = side["roofline"]["tick"]["achieved"] / bw["copy"]
            side["roofline"]["tick"]["frac_of_measured_triad"] = side["roofline"]["tick"]["achieved"] / bw["triad"]
            for k in side["roofline"].get("kernels", []):
                k["frac_of_measured_copy"] = k["achieved_gbs"] / bw["copy"]
                k["frac_of_measured_triad"] = k["achieved_gbs"] / bw["triad"]
        except Exception as exc:
            side["roofline"]["copy_bandwidth_error"] = str(exc)[:200]
    maps_checkpoint(args.workload)
    # The line goes out FIRST, as soon as it is complete: neither the driver's record (N = 1) nor its scaling curve (N > 1) may
    # depend on anything below.  The extras -- and for N > 1 the one RCCL exchange the path offers -- come after it and land in
    # the side file.
    if dist is not None:
        dist.barrier()
    if rank == 0:
        emit(compact_line(out))
        write_side(args, dict(side, line=out))
    # Everything below is optional and the line is out: a rank that falls out of step in an extra (an exception between two
    # barriers on one rank only) must not leave the job hanging at the others' barrier.  After --post-deadline seconds every
    # rank says so and leaves (exit 0: the record stands; the side file says what is missing).
    extras = []
    running = {"name": None}   # the extra (or "gather_pose") in progress: what a stall is blamed on

    def post_deadline():
        # status 4, not 0: a hang or a stalled resident session in an extra is a failure of that extra, and the record must say
        # which one (the line itself is out and stands)
        print("bench.py: rank %d: extras / gather did not finish within %.0f s (running: %s); leaving with status 4 (the line was printed)" % (
            rank, args.post_deadline, running["name"]), file=sys.stderr, flush=True)
        if rank == 0:
            try:
                write_side(args, dict(side, line=out, extra=list(extras), extras_incomplete=True, stalled_in=running["name"]))
            except Exception:
                pass
        os._exit(4)
    post_timer = threading.Timer(args.post_deadline, post_deadline)
    post_timer.daemon = True
    post_timer.start()
    # Extra workloads.  One GPU: the full list.  Several GPUs: one 10^6-targets-per-GPU workload per YAML motion model plus
    # the configs' per-GPU shares, every rank in lockstep (same barriers, max over ranks).
    extra_names = [e for e in (args.extra if world == 1 else args.extra_multi).split(",") if e]
    for name in extra_names:
        if name == args.workload:
            continue
        running["name"] = name
        try:
            if name.endswith("_replay"):
                # The same population with the batches' chains FREE-RUNNING inside recorded graph blocks (one branch per batch,
                # joined only at the end of a block): legal for a replay, where the measurements of later ticks are already
                # there, not for a live stream, whose ticks arrive in lockstep.  Each chain runs ahead for a few ticks on its
                # own batch -- 180 / 228 MB at 10^6 targets, which fits the Infinity Cache although the population does not
                # (rocprofv3 kernel trace: AV x5 alone, then AR x5 alone).  Never `value`.
                base = name[:-len("_replay")]
                r = run_mixed(te, torch, base, args.extra_steps, 8, dist, rank, world, launch_mode="graph", reps=3)
                r["name"] = name
                r["launch_mode"] = "graph: one branch per batch, free-running inside blocks of ticks (replay only)"
            elif name in MIXED:
                small = sum(n for _, n in MIXED[name][1]) <= SMALL
                r = run_mixed(te, torch, name, 512 if small else args.extra_steps, 64 if small else 8, dist, rank, world,
                              launch_mode="auto" if args.launch_mode in ("fused", "live") else args.launch_mode, reps=3)
            elif name.endswith("_live"):                    # the resident mode of a small batch (never `value`)
                wl = name[:-len("_live")]
                if wl in MIXED:
                    r = run_mixed(te, torch, wl, args.extra_steps, 64, dist, rank, world, launch_mode="live", reps=5)
                else:
                    r = run_workload(te, torch, wl, args.extra_steps, 64, 0, dist=dist, rank=rank, world=world, launch_mode="live", reps=5)
                r["name"] = name
            elif name.endswith("_strong") and name[:-len("_strong")] in MIXED:   # a mixed population split over the ranks, model by model
                wl = name[:-len("_strong")]
                small = sum(n for _, n in MIXED[wl][1]) // world <= SMALL
                r = run_mixed(te, torch, wl, 512 if small else args.extra_steps, 64 if small else 8, dist, rank, world,
                              launch_mode="auto" if args.launch_mode in ("fused", "live") else args.launch_mode, reps=3, split=world)
                r["name"] = name
                r["desc"] += " -- strong scaling: the population split over %d GPUs, every model evenly" % world
            elif name.endswith("_strong"):                  # strong scaling: the workload's targets split over the ranks
                wl = name[:-len("_strong")]
                per_rank = WORKLOADS[wl][3] // world
                r = run_workload(te, torch, wl, args.extra_steps if per_rank > SMALL else 512, 8, 0, targets=per_rank,
                                 dist=dist, rank=rank, world=world, launch_mode=args.launch_mode, reps=3)
                r["name"] = name
                r["desc"] += " -- strong scaling: %d targets in total, %d per GPU" % (per_rank * world, per_rank)
            else:
                small = WORKLOADS[name][3] <= SMALL
                r = run_workload(te, torch, name, (4096 if name in RINGS else 1024) if small else args.extra_steps, 64 if small else 8, 0,
                                 dist=dist, rank=rank, world=world, launch_mode=args.launch_mode, reps=3)
        except Exception as exc:    # an extra never breaks the line
            extras.append({"name": name, "error": str(exc)[:200]})
            torch.cuda.empty_cache()
            continue
        attach_traffic(r["kernels"], name[:-len("_replay")] if name.endswith("_replay") else name, traffic)
        tr = [k.get("traffic") for k in r["kernels"]]
        extras.append({k: r[k] for k in ("name", "dtype", "targets_per_gpu", "layout", "kernel", "cycles_per_s", "ms_per_step",
                                        "device_ms_per_step", "achieved_gbs", "algorithmic_bytes_per_cycle", "residency", "launch_mode")}
                      | ({"live": r["live"], "achieved_gbs": None} if "live" in r else {})   # a resident launch moves the measurements only: no per-tick HBM figure
                      | {"roofline_frac": None if "live" in r else r["achieved_gbs"] / HBM_PEAK_GBS, "n_gpus": world,
                         "traffic_per_step": (sum(tr) if all(t is not None for t in tr) else None)})
        torch.cuda.empty_cache()
        maps_checkpoint(name)
    # RCCL needs one GPU per rank: a gloo rehearsal (several ranks on ONE GPU) cannot create the communicator
    if not args.no_gather and (world == 1 or dist.get_backend() == "nccl"):
        running["name"] = "gather_pose"
        try:
            gr = gather_report(te, torch, dist, rank, world, deadline_s=args.gather_deadline)
            extras.append(gr)
            side["gather_pose_ms"] = {"exposed": gr["gather_pose_ms_exposed"], "overlapped": gr["gather_pose_ms_overlapped"],
                                      "rows_total": gr["rows_total"], "ticks_overlapped_with": gr["ticks"], "comm_ranks": gr["comm_ranks"]}
        except TimeoutError as exc:
            # a peer never arrived: say so and leave non-zero -- never re-exec, never wait for ever (the line is out already)
            print("bench.py: pose gather did not finish: %s" % exc, file=sys.stderr)
            if rank == 0:
                write_side(args, dict(side, line=out, extra=extras, gather_pose_error=str(exc)))
            sys.stderr.flush()
            os._exit(3)
        except Exception as exc:
            extras.append({"name": "gather_pose", "error": str(exc)[:300]})
    if extras:
        side["extra"] = extras
        # the HBM-bound rows (state > 1 GB) apart, where a reader finds them
        side["roofline"]["hbm_bound"] = {e["name"]: {"achieved": e["achieved_gbs"], "frac": e["roofline_frac"], "cycles_per_s": e["cycles_per_s"]}
                                         for e in extras if "error" not in e and e.get("residency", "").startswith("HBM-bound") and not e["name"].endswith("_replay") and "live" not in e}
        side["roofline"]["l3_assisted"] = {e["name"]: {"achieved": e["achieved_gbs"], "frac": e["roofline_frac"], "cycles_per_s": e["cycles_per_s"]}
                                           for e in extras if "error" not in e and e.get("residency", "").startswith("L3-assisted") and not e["name"].endswith("_replay") and "live" not in e}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    post_timer.cancel()
    if rank == 0 and world == 1:
        side["slower_than_previous_record"] = slowdown_check(
            dict({e["name"]: e["ms_per_step"] for e in extras if "ms_per_step" in e}, **{out["config"]["name"]: out["ms_per_step"]}))
    if rank == 0:
        write_side(args, dict(side, line=out))
        print(json.dumps(dict(side, line=out)), file=sys.stderr)


if __name__ == "__main__":
    main()
