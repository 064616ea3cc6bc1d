#!/usr/bin/env python3
"""bench.py -- predict+update cycles/s of the batched Kalman path on MI355X.

One "step" = one tick = one pass of the hot path (predict + measurement update of every target)
over one batch of synthetic measurements already resident in HBM.  Default workload: BASELINE.json
configs[1] (10 000 uniform-velocity targets, fp64, one GPU).  Other workloads via --workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4ar|cfg4av|uv1m|...]

For N > 1 the driver launches one rank per GPU with torch.distributed.run; targets are independent,
so each rank owns its own shard (weak scaling: the per-GPU workload is fixed) and there is no
data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL_P_BYTES = {"uniform_velocity": 91, "uniform_acceleration": 187, "angular_velocities": 325, "angular_rates": 697}  # words, SURVEY 8d
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

# lanes-per-target tuned on MI355X (tools/sweep.py): small, latency-bound batches want more lanes per
# target, large HBM-bound ones fewer.  0 = the library default.
# 0 = automatic: the shipped models' Q, R, P0 are symmetric and do not couple axes, so the library picks the
# axis-separable layout with symmetric-packed group blocks (301 = 1 + TARGET_LAYOUT_AXIS_SEPARABLE_PACKED).
# *_s201 force the axis-separable layout with full group blocks (bit-identical to the dense kernel, keeps the
# reference's rounding-level asymmetry of P); *_full / *_packed force the dense kernel (what general matrices
# get): full P, or symmetric-packed P (101 = 1 + TARGET_LAYOUT_SYMMETRIC_PACKED).
TUNED_LANES = {"cfg2_full": 3, "uv1m_full": 1, "ua1m_full": 1, "ar1m_full": 6, "av1m_full": 3, "uv1m_packed": 101,
               "ar1m_packed": 103, "av1m_packed": 101,   # what coupled but symmetric Q, R, P0 get automatically
               "uv1m_s201": 201, "ua1m_s201": 201, "av1m_s201": 201, "ar1m_s201": 201}

GRAPH_PASSES = 4   # passes over a one-block measurement ring per recorded graph (small batches only)
GRAPH_TICKS = 64   # ticks per block of the measurement ring = per recorded hipGraph (best block length for configs[1]: 16/32/64/128 -> 2.65/2.38/2.26/2.83 us per tick)
# measurement ring lengths (ticks) of the *_stream workloads: the ring is >= 1 GiB, far beyond L2 (32 MB) + Infinity Cache (256 MB)
RINGS = {"cfg2_stream": 2048, "cfg3_stream": 512}

# stream variants: availability < 1 = per-(target, tick) measurement mask (predict-only otherwise); rpy_noise = orientation noise
VARIANTS = {"ar1m_a90": dict(availability=0.9, rpy_noise=0.1), "av1m_a90": dict(availability=0.9, rpy_noise=0.1)}

WORKLOADS = {
    # name: (description, model, dtype, targets per GPU, seed)
    "cfg2": ("10000 targets, uniform-velocity model, fp64 (BASELINE.json configs[1])", "uniform_velocity", "f64", 10_000, 20240002),
    "cfg3": ("100000 targets, uniform-acceleration model, fp32 (configs[2])", "uniform_acceleration", "f32", 100_000, 20240003),
    "cfg2_stream": ("configs[1] with a measurement ring of 2048 ticks (1.1 GB): every tick's measurements come from HBM, not from L2",
                    "uniform_velocity", "f64", 10_000, 20240002),
    "cfg3_stream": ("configs[2] with a measurement ring of 512 ticks (1.4 GB)", "uniform_acceleration", "f32", 100_000, 20240003),
    "cfg4ar": ("angular-rates half of configs[3], 62500 targets per GPU, fp32", "angular_rates", "f32", 62_500, 20240004),
    "cfg4av": ("angular-velocities half of configs[3], 62500 targets per GPU, fp32", "angular_velocities", "f32", 62_500, 20240004),
    "uv1m": ("1000000 targets, uniform-velocity model, fp64", "uniform_velocity", "f64", 1_000_000, 20240012),
    "ua1m": ("1000000 targets, uniform-acceleration model, fp32", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "ar1m": ("1000000 targets, angular-rates model, fp32", "angular_rates", "f32", 1_000_000, 20240014),
    "av1m": ("1000000 targets, angular-velocities model, fp32", "angular_velocities", "f32", 1_000_000, 20240015),
    "cfg2_full": ("10000 targets, uniform-velocity model, fp64, dense kernel with full P", "uniform_velocity", "f64", 10_000, 20240002),
    "uv1m_full": ("1000000 targets, uniform-velocity model, fp64, dense kernel with full P", "uniform_velocity", "f64", 1_000_000, 20240012),
    "uv1m_packed": ("1000000 targets, uniform-velocity model, fp64, dense kernel with symmetric-packed P", "uniform_velocity", "f64", 1_000_000, 20240012),
    "ar1m_packed": ("1000000 targets, angular-rates model, fp32, dense kernel with symmetric-packed P, 3 lanes per target", "angular_rates", "f32", 1_000_000, 20240014),
    "av1m_packed": ("1000000 targets, angular-velocities model, fp32, dense kernel with symmetric-packed P", "angular_velocities", "f32", 1_000_000, 20240015),
    "ua1m_full": ("1000000 targets, uniform-acceleration model, fp32, dense kernel with full P", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "ar1m_full": ("1000000 targets, angular-rates model, fp32, dense kernel with full P", "angular_rates", "f32", 1_000_000, 20240014),
    "av1m_full": ("1000000 targets, angular-velocities model, fp32, dense kernel with full P", "angular_velocities", "f32", 1_000_000, 20240015),
    "uv1m_s201": ("1000000 targets, uniform-velocity model, fp64, axis-separable layout with full group blocks", "uniform_velocity", "f64", 1_000_000, 20240012),
    "ua1m_s201": ("1000000 targets, uniform-acceleration model, fp32, axis-separable layout with full group blocks", "uniform_acceleration", "f32", 1_000_000, 20240013),
    "av1m_s201": ("1000000 targets, angular-velocities model, fp32, axis-separable layout with full group blocks", "angular_velocities", "f32", 1_000_000, 20240015),
    "ar1m_s201": ("1000000 targets, angular-rates model, fp32, axis-separable layout with full group blocks", "angular_rates", "f32", 1_000_000, 20240014),
    "ar1m_a90": ("1000000 targets, angular-rates model, fp32, measurements on 90 % of the (target, tick) pairs, orientation noise 0.1 rad (SURVEY 8d variant)", "angular_rates", "f32", 1_000_000, 20240018),
    "av1m_a90": ("1000000 targets, angular-velocities model, fp32, measurements on 90 % of the (target, tick) pairs, orientation noise 0.1 rad (SURVEY 8d variant)", "angular_velocities", "f32", 1_000_000, 20240019),
    "uv1m32": ("1000000 targets, uniform-velocity model, fp32", "uniform_velocity", "f32", 1_000_000, 20240020),
    "ua1m64": ("1000000 targets, uniform-acceleration model, fp64", "uniform_acceleration", "f64", 1_000_000, 20240021),
    "ar1m64": ("1000000 targets, angular-rates model, fp64", "angular_rates", "f64", 1_000_000, 20240016),
    "av1m64": ("1000000 targets, angular-velocities model, fp64", "angular_velocities", "f64", 1_000_000, 20240017),
}


# BASELINE.json configs[3] / configs[4]: mixed populations, two batches per GPU, per-GPU share of 10^6 targets
# over 8 GPUs.  (name: list of (model, targets per GPU)); cfg5 adds the sphere-intersection query every tick.
MIXED = {
    "cfg4": ("configs[3]: 62500 angular-rates + 62500 angular-velocities targets per GPU (10^6 over 8 GPUs), fp32",
             [("angular_rates", 62_500), ("angular_velocities", 62_500)], "f32", 20240004, False),
    "cfg5": ("configs[4]: 62500 angular-rates + 62500 uniform-acceleration targets per GPU + sphere intersection every tick, fp32",
             [("angular_rates", 62_500), ("uniform_acceleration", 62_500)], "f32", 20240005, True),
}


def run_mixed(te, torch, name, steps, warmup, dist=None, rank=0, world=1, stream_ticks=64, scale=1, launch_mode="graph"):
    """Two batches in one manager, one step launch per batch per tick (+ one intersection launch per batch for
    cfg5); in graph mode the batches are concurrent branches of one hipGraph (target_manager_step_sequence_all).
    Returns cycles/s over both batches; algorithmic bytes are the sum of the batches' figures."""
    import numpy as np
    from target_estimation_amd.streams import make_stream
    desc, parts, dtype, seed, intersect = MIXED[name]
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = 1.0 / 250.0
    ticks = min(stream_ticks, steps + warmup)
    streams, base = [], 0
    for k, (model, n) in enumerate(parts):
        n *= scale
        mt = te.MODEL_TYPES[model]
        st = make_stream(mt, n, ticks, dt, seed + 1000 * rank + 17 * k)
        ids = np.arange(n, dtype=np.uint32) + base + rank * 10_000_000
        base += n
        params = _model_params(model)
        mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy(), None, None, type=mt, Q=params["Q"], R=params["R"], P0=params["P"])
        streams.append((mt, st))
    batches = mgr.batches()
    assert len(batches) == len(parts)
    meas = [st["meas"].to(b.torch_dtype()).contiguous() for (_, st), b in zip(streams, batches)]
    origin = np.zeros(3)
    outs = [(torch.empty(b.size, dtype=torch.float64, device="cuda"), torch.empty((b.size, 7), dtype=torch.float64, device="cuda"))
            for b in batches] if intersect else None
    lib = mgr._lib

    query = (origin, 1.0, [o[0] for o in outs], [o[1] for o in outs]) if intersect else None

    def tick(s):
        for j, b in enumerate(batches):
            b.step(dt, meas[j][s % ticks])
            if intersect:
                lib.target_batch_intersect_sphere_dev(b._h, float("nan"), origin.ctypes.data_as(te.capi.c_double_p), 1.0,
                                                      outs[j][0].data_ptr(), outs[j][1].data_ptr())

    def run(first, count):
        """`count` ticks starting at stream position `first`: whole passes over the stream replay ONE hipGraph whose
        branches are the batches (they run concurrently); the rest is issued eagerly."""
        if launch_mode == "python":
            for s in range(first, first + count):
                tick(s)
            return
        s, end = first, first + count
        while s < end:
            o = s % ticks
            if passes > 1 and o == 0 and end - s >= ticks * passes:   # several passes over the ring in one graph
                mgr.step_sequence_all(dt, meas, query=query, use_graph=1, n_ticks=ticks * passes)
                s += ticks * passes
                continue
            nblk = min(ticks - o, end - s)
            whole = launch_mode == "graph" and o == 0 and nblk == ticks
            mgr.step_sequence_all(dt, [m[o:o + nblk] for m in meas], query=query, use_graph=1 if whole else 0)
            s += nblk

    passes = GRAPH_PASSES if (launch_mode == "graph" and ticks == GRAPH_TICKS and scale == 1) else 1
    if launch_mode == "graph":
        mgr.step_sequence_all(dt, meas, query=query, use_graph=2)    # record before the timed region
        if passes > 1:
            mgr.step_sequence_all(dt, meas, query=query, use_graph=2, n_ticks=ticks * passes)
    run(0, warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(warmup, steps)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    dev_ms = ev0.elapsed_time(ev1)
    n_total = sum(b.size for b in batches)
    alg = sum(b.algorithmic_bytes * b.size for b in batches)
    if intersect:
        alg += sum(8 * 8 * b.size for b in batches)   # query: writes delta + pose7 (doubles); its input is the step's state
    for b in batches:
        p, _, _ = b.get_est(twist=False, acc=False)
        assert torch.isfinite(p).all()
    res = dict(name=name, desc=desc, model="+".join(m for m, _ in parts), dtype=dtype, targets_per_gpu=n_total,
               lanes_per_target=1, layout="+".join(b.layout for b in batches), elapsed_s=elapsed, ms_per_step=elapsed * 1e3 / steps,
               cycles_per_s=n_total * world * steps / elapsed, device_ms_per_launch=dev_ms / steps,
               algorithmic_bytes_per_cycle=alg / n_total, algorithmic_bytes_per_launch=alg,
               achieved_gbs=alg / (dev_ms * 1e-3 / steps) / 1e9, resident_bytes_per_target=0.0,
               launch_mode=launch_mode)
    if intersect:
        hit = sum(int((o[0] > -1).sum()) for o in outs)
        res["intersections_last_tick"] = hit
    mgr.close()
    return res


def _model_params(model):
    """Q, R, P0 of a shipped model file (row-major), through the oracle's YAML reader-independent path."""
    import yaml
    import numpy as np
    with open(os.path.join(ROOT, "models", "model_%s_params.yaml" % model)) as f:
        node = yaml.safe_load(f)
    n = {"uniform_velocity": 6, "uniform_acceleration": 9, "angular_velocities": 12, "angular_rates": 18}[model]
    m = 3 if n in (6, 9) else 6
    return dict(Q=np.array(node["Q"]).reshape(n, n), R=np.array(node["R"]).reshape(m, m), P=np.array(node["P"]).reshape(n, n))


def run_workload(te, torch, name, steps, warmup, lanes, targets=None, dist=None, rank=0, world=1, stream_ticks=64,
                 launch_mode="graph", gather=False):
    from target_estimation_amd.streams import make_stream
    desc, model, dtype, n_targets, seed = WORKLOADS[name]
    if targets:
        n_targets = targets
    if not lanes:
        lanes = TUNED_LANES.get(name, 0)
    path = os.path.join(ROOT, "models", "model_%s_params.yaml" % model)
    mgr = te.TargetManager(path, dtype=dtype, lanes_per_target=lanes)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    mtype = te.MODEL_TYPES[model]
    dt = 1.0 / 250.0
    # The synthetic measurements live in HBM as a ring of `ticks` ticks that the run cycles through; in graph mode
    # every GRAPH_TICKS-tick block of the ring is one recorded hipGraph.  Default ring = one block; RINGS gives
    # the *_stream workloads a ring far larger than L2 + Infinity Cache, so that every tick's measurements come
    # from HBM itself.
    # The block length adapts to --steps so that the timed region is whole blocks for any K: the largest
    # divisor of K in [16, GRAPH_TICKS], else GRAPH_TICKS (the remainder is then enqueued launch by launch).
    gb = GRAPH_TICKS
    if steps % GRAPH_TICKS:
        gb = next((g for g in range(min(GRAPH_TICKS, steps), 15, -1) if steps % g == 0), min(GRAPH_TICKS, steps))
    ring_want = RINGS.get(name, stream_ticks)
    ticks = max(gb, min(ring_want, steps + warmup) // gb * gb)
    st = make_stream(mtype, n_targets, ticks, dt, seed + 1000 * rank, **VARIANTS.get(name, {}))
    has = st["has_meas"]    # [ticks, N] uint8 or None
    import numpy as np
    ids = np.arange(n_targets, dtype=np.uint32) + rank * n_targets  # global ids: rank-contiguous shards
    mgr.init_batch(ids, dt, 0.0, st["p0"].cpu().numpy())
    b = mgr.batches()[0]
    meas = st["meas"].to(b.torch_dtype()).contiguous()   # [ticks, 7, N] in the batch precision
    del st
    torch.cuda.synchronize()
    # A one-block ring of a small (launch-bound) batch is replayed GRAPH_PASSES times per recorded graph
    # (target_batch_step_sequence_ring): every graph launch costs ~8 us, a noticeable share of 64 x 2 us.
    passes = 1
    if launch_mode == "graph" and ticks == gb and n_targets <= 200000:
        passes = next((q for q in range(GRAPH_PASSES, 0, -1) if steps % (gb * q) == 0), 1)

    # One launch of the step kernel per tick in every mode.  "python": one C-ABI call per tick;
    # "sequence": the launches of a block of ticks are enqueued by one C call
    # (target_batch_step_sequence); "graph": that block is a recorded hipGraph that is replayed.
    done = [0]

    def run_ticks(count):
        if launch_mode == "python":
            for _ in range(count):
                b.step(dt, meas[done[0] % ticks], None if has is None else has[done[0] % ticks])
                done[0] += 1
            return
        while count > 0:
            off = done[0] % ticks
            if passes > 1 and off == 0 and count >= gb * passes:      # several passes over the ring in one graph
                b.step_sequence(dt, meas, has, use_graph=True, n_ticks=gb * passes)
                done[0] += gb * passes
                count -= gb * passes
                continue
            blk = min(count, gb - off % gb)
            # only whole blocks are replayed from the recorded graphs (recorded before the timed region);
            # partial blocks are enqueued launch by launch
            if launch_mode == "fused":   # temporally fused: the whole block in ONE launch ("effective" metric)
                b.step_fused(dt, meas[off:off + blk], None if has is None else has[off:off + blk])
            else:
                b.step_sequence(dt, meas[off:off + blk], None if has is None else has[off:off + blk],
                                use_graph=(launch_mode == "graph" and off % gb == 0 and blk == gb))
            done[0] += blk
            count -= blk

    if launch_mode == "graph":
        for off in range(0, ticks - gb + 1, gb):   # record every block's graph now (set-up; launches nothing)
            b.step_sequence(dt, meas[off:off + gb], None if has is None else has[off:off + gb], use_graph=2)
        if passes > 1:
            b.step_sequence(dt, meas, has, use_graph=2, n_ticks=gb * passes)
    run_ticks(warmup)
    done[0] = 0          # the timed region starts on a block boundary of the ring
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_ticks(steps)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    dev_ms = ev0.elapsed_time(ev1)
    # sanity: the state is finite after the run
    x, P = mgr.get_state_batch(ids[:64])
    assert np.isfinite(x).all() and np.isfinite(P).all()
    launch_s = dev_ms * 1e-3 / steps
    alg_bytes = b.algorithmic_bytes * n_targets
    res = dict(
        name=name, desc=desc, model=model, dtype=dtype, targets_per_gpu=n_targets,
        lanes_per_target=b.lanes_per_target, layout=b.layout, elapsed_s=elapsed, ms_per_step=elapsed * 1e3 / steps,
        cycles_per_s=n_targets * world * steps / elapsed,
        device_ms_per_launch=launch_s * 1e3,
        algorithmic_bytes_per_cycle=b.algorithmic_bytes, algorithmic_bytes_per_launch=alg_bytes,
        achieved_gbs=alg_bytes / launch_s / 1e9,
        resident_bytes_per_target=b.resident_bytes_per_target, launch_mode=launch_mode,
        measurement_ring_ticks=ticks, measurement_ring_bytes=int(meas.numel() * meas.element_size()))
    if gather and dist is not None:
        from target_estimation_amd import dist as td
        pose, _, _ = b.get_est(twist=False, acc=False)
        if dist.get_backend() != "nccl":
            pose = pose.cpu()
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        full = td.gather_rows(pose, n_targets * world, dst=0)
        torch.cuda.synchronize()
        res["gather_pose_ms"] = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert full.shape == (n_targets * world, 7) and bool(torch.isfinite(full).all())
    res["_mgr"] = (mgr, b, None, ids, dt)
    return res


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


def launch_floor(torch, nbytes, reps=256, rounds=8):
    """Period of a DEPENDENT launch on this box: a trivial in-place read-modify-write of `nbytes` (the tick's
    working set), `reps` launches recorded in a graph and replayed.  No tick over that working set can be
    shorter with one launch per tick; it bounds the roofline fraction of launch-bound (small) batches."""
    buf = torch.zeros(max(1, nbytes // 8), dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            buf.mul_(1.0000001)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            buf.mul_(1.0000001)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * rounds)


def cpu_baseline(name, targets=None, budget_s=6.0):
    """The CPU oracle (oracle/, the 'port' of the reference's Eigen path) with OpenMP over targets
    on this box's host cores, on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    desc, model, dtype, n_targets, seed = WORKLOADS[name]
    if targets:
        n_targets = targets
    n_cpu = min(n_targets, 20000)
    m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_%s_params.yaml" % model))
    rng = np.random.default_rng(seed)
    p0 = np.concatenate([rng.uniform(-10, 10, (n_cpu, 3)), np.tile([0, 0, 0, 1.0], (n_cpu, 1))], 1)
    dt = 1.0 / 250.0
    ob = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype, fast=True)
    threads = host_threads(oracle.load(True).orc_max_threads())
    meas = p0.copy()
    ob.step(dt, meas, nthreads=threads)  # warm
    t0 = time.perf_counter()
    ticks = 0
    while True:
        meas[:, :3] += 0.004 + rng.normal(0, 0.01, (n_cpu, 3))
        ob.step(dt, meas, nthreads=threads)
        ticks += 1
        if time.perf_counter() - t0 > budget_s and ticks >= 5:
            break
    el = time.perf_counter() - t0
    # one-thread figure on a shorter sample (SURVEY 8d asks for both)
    t1 = time.perf_counter()
    ticks1 = 0
    while time.perf_counter() - t1 < 1.5 or ticks1 < 2:
        ob.step(dt, meas, nthreads=1)
        ticks1 += 1
    el1 = time.perf_counter() - t1
    return dict(value=n_cpu * ticks / el, unit="cycles/s", cores=int(threads), kind="port",
                sample="%d %s targets x %d ticks (%s, OpenMP static over targets, %.1f s)" % (n_cpu, model, ticks, dtype, el),
                value_1thread=n_cpu * ticks1 / el1)


def parity_report(te, torch, name, n_sample=256, checkpoints=(1, 100, 1000)):
    """parity next to the perf number.  A sample of the same workload (same model, precision, layout
    choice and stream generator) stepped on the GPU and by the CPU oracle; errors after 1 / 100 / 1000 ticks.
    The oracle is the checker here, nothing of it is timed."""
    import numpy as np
    import oracle
    from target_estimation_amd.streams import make_stream
    desc, model, dtype, _, seed = WORKLOADS[name]
    m = oracle.load_model_yaml(os.path.join(ROOT, "models", "model_%s_params.yaml" % model))
    dt, ticks = 1.0 / 250.0, max(checkpoints)
    st = make_stream(te.MODEL_TYPES[model], n_sample, ticks, dt, seed)
    p0 = st["p0"].cpu().numpy()
    ids = np.arange(n_sample, dtype=np.uint32)
    mgr = te.TargetManager(os.path.join(ROOT, "models", "model_%s_params.yaml" % model), dtype=dtype, lanes_per_target=TUNED_LANES.get(name, 0))
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    mgr.init_batch(ids, dt, 0.0, p0)
    b = mgr.batches()[0]
    meas = st["meas"].to(b.torch_dtype()).contiguous()
    meas_host = meas.to(torch.float64).cpu().numpy()           # the oracle sees what the kernel saw
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype)
    rep = dict(targets=n_sample, layout=b.layout, ticks=list(checkpoints), max_abs_x=[], max_rel_x=[], max_rel_P=[],
               ids_exact=bool((b.slot_ids() == ids).all()), oracle="oracle/te_oracle.c, same precision (%s)" % dtype,
               tolerance=("x: 1e-10 + 1e-10|x|, P: 1e-9 max|P|" if dtype == "f64" else "x: 2e-3 + 1e-4|x|, P: 2e-3 max|P|"))
    # an fp32 batch is also compared with the fp64 oracle: the end-to-end precision loss
    orc64 = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype="f64") if dtype == "f32" else None
    if orc64 is not None:
        rep["vs_f64_oracle"] = dict(max_abs_x=[], max_rel_P=[])
    for s in range(ticks):
        b.step(dt, meas[s])
        row = np.ascontiguousarray(meas_host[s].T)
        orc.step(dt, row)
        if orc64 is not None:
            orc64.step(dt, row)
        if s + 1 in checkpoints:
            x, P = mgr.get_state_batch(ids)
            xo, Po = orc.state()
            rep["max_abs_x"].append(float(np.abs(x - xo).max()))
            rep["max_rel_x"].append(float((np.abs(x - xo) / np.maximum(np.abs(xo), 1e-3)).max()))
            rep["max_rel_P"].append(float((np.abs(P - Po) / np.abs(Po).max(axis=(1, 2), keepdims=True)).max()))
            if orc64 is not None:
                x64, P64 = orc64.state()
                rep["vs_f64_oracle"]["max_abs_x"].append(float(np.abs(x - x64).max()))
                rep["vs_f64_oracle"]["max_rel_P"].append(float((np.abs(P - P64) / np.abs(P64).max(axis=(1, 2), keepdims=True)).max()))
    mgr.close()
    return rep


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)   # whole 64-tick graph blocks
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + sorted(MIXED))
    ap.add_argument("--lanes", type=int, default=0, help="lanes per target (0 = tuned default)")
    ap.add_argument("--targets", type=int, default=0, help="override targets per GPU")
    ap.add_argument("--extra", default="cfg2_stream,cfg3,cfg3_stream,cfg4,cfg5,cfg4x8,cfg5x8,uv1m,ua1m,av1m,ar1m,ar1m_a90,av1m_a90,ar1m_s201,av1m_s201,cfg2_full,uv1m_full,uv1m_packed,ar1m_full,ar1m_packed", help="comma list of extra workloads reported under 'extra' (N=1 only; '' = none)")
    ap.add_argument("--extra-multi", default="uv1m,ua1m,av1m,ar1m,cfg4,cfg5,uv1m_strong,ar1m_strong",
                    help="extra workloads when --gpus > 1 (per-GPU sizes; every rank runs them in lockstep; NAME_strong = the workload's "
                         "targets split over the ranks)")
    ap.add_argument("--extra-steps", type=int, default=50)
    ap.add_argument("--stream-ticks", type=int, default=0, help="ticks of synthetic measurements kept in HBM and replayed (= ticks per recorded graph); 0 = default")
    ap.add_argument("--scale", type=int, default=1, help="mixed workloads (cfg4/cfg5): multiply the per-GPU populations (8 = all 10^6 targets on one GPU)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region, gather every rank's pose7 rows to rank 0 (RCCL gather over xGMI) and report its time; "
                         "off by default: the predict/update path has no collective")
    ap.add_argument("--launch-mode", default="graph", choices=["python", "sequence", "graph", "fused"],
                    help="how the per-tick launches are enqueued (always one kernel launch per tick)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        # "nccl" is RCCL on ROCm.  TE_BENCH_BACKEND=gloo rehearses the multi-rank path on a one-GPU box.
        dist_mod.init_process_group(os.environ.get("TE_BENCH_BACKEND", "nccl"))
        dist = dist_mod
    import target_estimation_amd as te

    if args.workload in MIXED:
        res = run_mixed(te, torch, args.workload, args.steps, args.warmup, dist, rank, world, scale=args.scale,
                        launch_mode="graph" if args.launch_mode == "fused" else args.launch_mode,
                        **({"stream_ticks": args.stream_ticks} if args.stream_ticks else {}))
        mgr = None
    else:
        res = run_workload(te, torch, args.workload, args.steps, args.warmup, args.lanes, args.targets or None,
                           dist, rank, world, launch_mode=args.launch_mode, gather=args.gather,
                           **({"stream_ticks": args.stream_ticks} if args.stream_ticks else {}))
        mgr = res.pop("_mgr")
    out = {
        "metric": "KF predict+update cycles/sec over N targets",
        "value": res["cycles_per_s"], "unit": "cycles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic",
        "config": {"workload": res["desc"], "name": res["name"], "motion_model": res["model"],
                   "targets_per_gpu": res["targets_per_gpu"], "targets_total": res["targets_per_gpu"] * world,
                   "lanes_per_target": res["lanes_per_target"], "P_layout": res["layout"], "dt": 0.004, "launch_mode": res["launch_mode"],
                   "measurements": ("synthetic, resident in HBM as a ring of %d ticks (%d MB) that the run cycles through; 'extra' has the same "
                                    "workload with a ring beyond L2 + Infinity Cache (cfg2_stream)" % (res["measurement_ring_ticks"], res["measurement_ring_bytes"] // 1000000))
                   if "measurement_ring_ticks" in res else "synthetic, resident in HBM",
                   "sharding": "contiguous id ranges per rank, no data-path collective"} |
                  ({"gather_pose_ms": res["gather_pose_ms"]} if "gather_pose_ms" in res else {}),
        "roofline": {"bound": "hbm", "achieved": res["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": res["achieved_gbs"] / HBM_PEAK_GBS, "traffic": None,
                     "kernel": ("kf_step_sep_kernel<%s,%s>" % (res["model"], res["dtype"]) if res["layout"].startswith("axis_separable")
                                else "kf_step_kernel<%s,%s,G=%d,%s>" % (res["model"], res["dtype"], res["lanes_per_target"], res["layout"])),
                     "survey_full_P_bytes_per_cycle": (FULL_P_BYTES[res["model"]] * (8 if res["dtype"] == "f64" else 4)
                                                       if res["model"] in FULL_P_BYTES else None),
                     "algorithmic_bytes_per_cycle": res["algorithmic_bytes_per_cycle"],
                     "algorithmic_bytes_per_launch": res["algorithmic_bytes_per_launch"],
                     "device_ms_per_launch": res["device_ms_per_launch"]},
    }
    del mgr
    # HBM bytes per launch from the committed rocprofv3 PMC passes (cannot be read inside this process)
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(args.workload)
        if tr and not args.targets and not args.lanes:
            out["roofline"]["traffic"] = tr["hbm_read_bytes"] + tr["hbm_write_bytes"]
            out["roofline"]["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
    except (OSError, ValueError):
        pass
    if world == 1 and rank == 0 and res["targets_per_gpu"] <= 200000:
        try:
            floor_s = launch_floor(torch, int(res["algorithmic_bytes_per_launch"] // 2))
            rf = out["roofline"]
            rf["launch_floor_ms"] = floor_s * 1e3
            rf["launch_floor_note"] = ("period of a trivial dependent read-modify-write launch over the same working set, graph-replayed; "
                                       "a launch-bound tick cannot beat it")
            rf["frac_at_launch_floor"] = res["algorithmic_bytes_per_launch"] / floor_s / 1e9 / HBM_PEAK_GBS
        except Exception as exc:   # never let the diagnostic break the bench line
            out["roofline"]["launch_floor_error"] = str(exc)[:200]
    if world == 1 and rank == 0 and not args.no_cpu and args.workload in WORKLOADS:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.targets or None)
        out["cpu_baseline"]["cpu_model"] = cpu_model()
        try:
            out["parity"] = parity_report(te, torch, args.workload)
        except Exception as exc:   # a diagnostic: never let it break the bench line
            out["parity"] = {"error": str(exc)[:300]}
    if world == 1 and rank == 0:
        try:
            bw = copy_bandwidth(torch)
            out["roofline"]["measured_copy_gbs"] = bw["copy"]
            out["roofline"]["measured_triad_gbs"] = bw["triad"]
            out["roofline"]["frac_of_measured_stream"] = res["achieved_gbs"] / max(bw.values())
            out["roofline"]["measured_stream_note"] = ("torch dst.copy_(src) and torch.add(a, b, out=c) over 1 GiB fp32 arrays on this box; "
                                                       "frac_of_measured_stream uses the larger of the two")
        except Exception as exc:
            out["roofline"]["copy_bandwidth_error"] = str(exc)[:200]
    # Extra workloads in the same line.  One GPU: the full list.  Several GPUs: one 10^6-targets-per-GPU
    # workload per YAML motion model plus configs[3], every rank in lockstep (same barriers, max over
    # ranks), so that the scaling curve exists for each model and not only for the headline workload.
    extra_names = [e for e in args.extra.split(",") if e]
    if world > 1:
        extra_names = [e for e in args.extra_multi.split(",") if e]
    extras = []
    for name in extra_names:
        if name == args.workload:
            continue
        base, _, mult = name.partition("x")            # "cfg5x8": 8 x the per-GPU share = all 10^6 targets on this GPU
        if base in MIXED:
            sc = int(mult) if mult else 1
            r = run_mixed(te, torch, base, 2048 if sc == 1 else 128, 256 if sc == 1 else 64, dist, rank, world, scale=sc,
                          launch_mode="graph" if args.launch_mode == "fused" else args.launch_mode)
            r["name"] = name
            if sc > 1:
                r["desc"] += " -- x%d: the whole population on this GPU" % sc
        elif name.endswith("_strong"):                  # strong scaling: the workload's targets split over the ranks
            wl = name[:-len("_strong")]
            per_rank = WORKLOADS[wl][3] // world
            r = run_workload(te, torch, wl, args.extra_steps if per_rank > 200000 else 640, 10, 0, targets=per_rank,
                             dist=dist, rank=rank, world=world, launch_mode=args.launch_mode)
            r.pop("_mgr")
            r["name"] = name
            r["desc"] += " -- strong scaling: %d targets in total, %d per GPU" % (per_rank * world, per_rank)
        else:
            small = WORKLOADS[name][3] <= 200000
            r = run_workload(te, torch, name, (4096 if name in RINGS else 1920) if small else args.extra_steps, 64 if small else 10, 0,
                             dist=dist, rank=rank, world=world, launch_mode=args.launch_mode)
            r.pop("_mgr")
        extras.append({k: r[k] for k in ("name", "desc", "dtype", "targets_per_gpu", "lanes_per_target", "layout", "cycles_per_s",
                                        "ms_per_step", "device_ms_per_launch", "achieved_gbs",
                                        "algorithmic_bytes_per_cycle")} | {"roofline_frac": r["achieved_gbs"] / HBM_PEAK_GBS,
                                                                            "n_gpus": world})
        torch.cuda.empty_cache()
    if extras:
        out["extra"] = extras
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
