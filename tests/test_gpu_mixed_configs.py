"""BASELINE.json configs[3] / configs[4] at their per-GPU size (62 500 + 62 500 targets, two motion models in one
manager) through target_manager_step_sequence_all -- the call bench.py times -- against the ORACLE on a 2 000-target
sample per model: state and covariance, and for configs[4] the fused per-tick sphere query's delta.  (Round 1 only
compared this path with the library's own per-batch calls.)

Round 3: the same call at the HEADLINE's own size and schedule -- bench.py's `value` is cfg4_1gpu = 500 000 + 500 000 fp64
targets through the EAGER target_manager_step_sequence_all, whose all-batches zig-zag (batch order and tile order reversed
on odd ticks, csrc/target_manager.cpp stepSequenceAll; reference semantics: the caller's loop over every target every
tick, src/target_manager.cpp:190-225) is only taken from 128 MB of state.  The oracle regenerates the measurement stream
itself (oracle.stream_sample, the CPU twin of the library's keyed generator): nothing is copied back but the results."""
import os
import subprocess
import sys
import numpy as np
import pytest

import oracle
from test_gpu_parity import check_state

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def _build(models, parts, dtype, ticks, dt, seed, accel_scene=False):
    from target_estimation_amd.streams import make_stream
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    base, out = 0, []
    for k, (name, n) in enumerate(parts):
        m = models[name]
        st = make_stream(te.MODEL_TYPES[name], n, ticks, dt, seed + 17 * k, dtype=dtype)
        ids = np.arange(n, dtype=np.uint32) + base
        base += n
        p0 = st["p0"].cpu().numpy()
        v0 = a0 = None
        if accel_scene:   # inbound, accelerating targets so that the sphere query has roots to find
            rng = np.random.default_rng(seed + k)
            d = p0[:, :3] / np.linalg.norm(p0[:, :3], axis=1, keepdims=True)
            v0 = np.concatenate([-d * rng.uniform(1, 6, (n, 1)), np.zeros((n, 3))], 1)
            a0 = np.concatenate([rng.normal(0, 1.0, (n, 3)) + [0, 0, -2.0], np.zeros((n, 3))], 1)
        assert mgr.init_batch(ids, dt, 0.0, p0, v0, a0, type=te.MODEL_TYPES[name], Q=m["Q"], R=m["R"], P0=m["P"]) == n
        out.append(dict(name=name, ids=ids, p0=p0, v0=v0, a0=a0, meas=st["meas"], seed=seed + 17 * k))
    batches = mgr.batches()
    assert len(batches) == len(parts)
    meas = [o["meas"] for o in out]
    assert all(m.dtype == b.torch_dtype() for m, b in zip(meas, batches))
    return mgr, batches, out, meas


def _oracle_on_sample(models, o, sample, ticks, dt, dtype):
    """The oracle over `sample` (indices into the batch) for `ticks` ticks, on the stream it regenerates itself."""
    m = models[o["name"]]
    ref = oracle.stream_sample(m["model"], o["seed"], sample, ticks, dt, dtype=dtype)
    np.testing.assert_array_equal(ref["p0"][:, :3], o["p0"][sample][:, :3])     # same keyed stream on both sides
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], o["p0"][sample], dt, 0.0,
                             None if o["v0"] is None else o["v0"][sample], None if o["a0"] is None else o["a0"][sample], dtype=dtype)
    for s in range(ticks):
        orc.step(dt, ref["meas"][s])
    return orc


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("use_graph", [1, 0])
def test_config3_share_matches_oracle(models, dtype, use_graph):
    """62 500 angular-rates + 62 500 angular-velocities targets (one GPU's share of configs[3])."""
    import bench
    parts = bench.MIXED["cfg4"][1]
    ticks, dt = 8, 0.004
    mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, 20240004)
    half = ticks // 2
    for part in (slice(0, half), slice(half, ticks)):
        mgr.step_sequence_all(dt, [m[part] for m in meas], use_graph=use_graph)
    torch.cuda.synchronize()
    for o, b, mm in zip(info, batches, meas):
        m = models[o["name"]]
        n = len(o["ids"])
        sample = np.sort(np.random.default_rng(1).choice(n, 2000, replace=False))
        orc = _oracle_on_sample(models, o, sample, ticks, dt, dtype)
        check_state(mgr, o["ids"][sample], orc, dtype, "%s sample of configs[3]" % o["name"])
        np.testing.assert_array_equal(b.slot_ids()[::997], o["ids"][::997])
        assert mgr.getNumberMeasurements(int(o["ids"][-1])) == ticks
    mgr.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_config4_share_with_fused_query_matches_oracle(models, dtype):
    """62 500 angular-rates + 62 500 uniform-acceleration targets with the own-time sphere query of every target
    inside the step kernels (configs[4]): state AND delta of a 2 000-target sample against the oracle."""
    import bench
    parts = bench.MIXED["cfg5"][1]
    ticks, dt = 6, 0.004
    origin, radius = np.zeros(3), 5.0
    mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, 20240005, accel_scene=True)
    deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in batches]
    poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in batches]
    mgr.step_sequence_all(dt, meas, query=(origin, radius, deltas, poses), use_graph=1)
    torch.cuda.synchronize()
    n_hit = 0
    for o, b, mm, dd, pp in zip(info, batches, meas, deltas, poses):
        m = models[o["name"]]
        n = len(o["ids"])
        sample = np.sort(np.random.default_rng(2).choice(n, 2000, replace=False))
        orc = _oracle_on_sample(models, o, sample, ticks, dt, dtype)
        check_state(mgr, o["ids"][sample], orc, dtype, "%s sample of configs[4]" % o["name"])
        ok_o, pose_o, delta_o = orc.intersection_pose(ticks * dt, origin, radius)
        d = dd.cpu().numpy()[sample]
        hit, hit_o = d > -1, delta_o > -1
        assert (hit != hit_o).mean() <= (0.0 if dtype == "f64" else 0.01)
        both = hit & hit_o
        rtol = 1e-8 if dtype == "f64" else 5e-4
        np.testing.assert_allclose(d[both], delta_o[both], rtol=rtol, atol=rtol)
        np.testing.assert_allclose(pp.cpu().numpy()[sample][both], pose_o[both], atol=1e-7 if dtype == "f64" else 1e-2)
        n_hit += int(both.sum())
    assert n_hit > 200       # the scene has intersections to find
    mgr.close()


def _check_query(dd, pp, sample, orc, t_now, origin, radius, dtype):
    ok_o, pose_o, delta_o = orc.intersection_pose(t_now, origin, radius)
    d = dd.cpu().numpy()[sample]
    hit, hit_o = d > -1, delta_o > -1
    assert (hit != hit_o).mean() <= (0.0 if dtype == "f64" else 0.01)
    both = hit & hit_o
    rtol = 1e-8 if dtype == "f64" else 5e-4
    np.testing.assert_allclose(d[both], delta_o[both], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(pp.cpu().numpy()[sample][both], pose_o[both], atol=1e-7 if dtype == "f64" else 1e-2)
    return int(both.sum())


@pytest.mark.parametrize("wl", ["cfg4_1gpu", "cfg5_1gpu", "cfg5_1gpu64"])
def test_headline_path_at_full_size_matches_oracle(models, wl):
    """bench.py's headline workload as bench.py runs it: 10^6 targets in two batches, ONE eager
    target_manager_step_sequence_all call for an odd number of ticks (5), so that both traversal directions and both batch
    orders of the manager-level zig-zag run; 2 000-target sample per model against the oracle (state, covariance, and for
    configs[4] the fused query's delta / pose after the last tick)."""
    import bench
    desc, parts, dtype, seed, intersect = bench.MIXED[wl]
    ticks, dt = 5, 0.004
    origin, radius = np.zeros(3), 5.0
    mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, seed, accel_scene=intersect)
    state = sum(b.resident_bytes_per_target * b.size for b in batches)
    assert state >= 128 << 20, "below the zig-zag threshold: this test would not cover the headline's branch"
    query = None
    if intersect:
        deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in batches]
        poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in batches]
        query = (origin, radius, deltas, poses)
    mgr.step_sequence_all(dt, meas, query=query, use_graph=0)          # the bench's call: eager, all ticks in one call
    torch.cuda.synchronize()
    n_hit = 0
    for j, (o, b) in enumerate(zip(info, batches)):
        n = len(o["ids"])
        sample = np.sort(np.random.default_rng(3 + j).choice(n, 2000, replace=False))
        sample[0], sample[-1] = 0, n - 1                                 # first and last tile in both directions
        orc = _oracle_on_sample(models, o, sample, ticks, dt, dtype)
        check_state(mgr, o["ids"][sample], orc, dtype, "%s sample of %s" % (o["name"], wl))
        if intersect:
            n_hit += _check_query(deltas[j], poses[j], sample, orc, ticks * dt, origin, radius, dtype)
        np.testing.assert_array_equal(b.slot_ids()[::9973], o["ids"][::9973])
        assert mgr.getNumberMeasurements(int(o["ids"][-1])) == ticks and mgr.getNumberMeasurements(int(o["ids"][0])) == ticks
    if intersect:
        assert n_hit > 200
    # a second call continues the alternation where the first one stopped (odd tick count): still the oracle's result
    mgr.step_sequence_all(dt, [m[:2] for m in meas], query=query, use_graph=0)
    torch.cuda.synchronize()
    o, b = info[0], batches[0]
    sample = np.arange(0, len(o["ids"]), len(o["ids"]) // 500)[:500]
    m = models[o["name"]]
    ref = oracle.stream_sample(m["model"], o["seed"], sample, ticks, dt, dtype=dtype)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], o["p0"][sample], dt, 0.0,
                             None if o["v0"] is None else o["v0"][sample], None if o["a0"] is None else o["a0"][sample], dtype=dtype)
    for s in list(range(ticks)) + [0, 1]:
        orc.step(dt, ref["meas"][s])
    check_state(mgr, o["ids"][sample], orc, dtype, "%s after the second call of %s" % (o["name"], wl))
    mgr.close()


def _zigzag_small_case():
    """Run in a child process with TE_ZIGZAG_MIN_MB=0 (the threshold is read once per process): three small batches, ragged
    last tiles, the manager-level zig-zag (eager all-batches call) and the per-batch zig-zag (step_sequence), against the
    oracle on EVERY target."""
    import yaml  # noqa: F401
    from conftest import MODEL_FILES, model_path
    models = {k: oracle.load_model_yaml(model_path(k)) for k in MODEL_FILES}
    assert os.environ.get("TE_ZIGZAG_MIN_MB") == "0"
    for dtype in ("f64", "f32"):
        parts = [("uniform_velocity", 777), ("angular_rates", 333), ("uniform_acceleration", 1501)]
        ticks, dt = 7, 0.004
        mgr, batches, info, meas = _build(models, parts, dtype, ticks, dt, 4242)
        mgr.step_sequence_all(dt, [m[:5] for m in meas], use_graph=0)                       # manager-level zig-zag, odd count
        for b, m in zip(batches, meas):
            b.step_sequence(dt, m[5:7], None, use_graph=False)                               # per-batch zig-zag
        torch.cuda.synchronize()
        for o in info:
            sample = np.arange(len(o["ids"]))
            orc = _oracle_on_sample(models, o, sample, ticks, dt, dtype)
            check_state(mgr, o["ids"], orc, dtype, "%s, zig-zag forced on a small batch" % o["name"])
        mgr.close()
    print("zigzag small case ok")


def test_zigzag_forced_on_small_batches_matches_oracle():
    env = dict(os.environ, TE_ZIGZAG_MIN_MB="0", PYTHONPATH=os.pathsep.join([os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__))]))
    p = subprocess.run([sys.executable, "-c", "import test_gpu_mixed_configs as t; t._zigzag_small_case()"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "zigzag small case ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def _pingpong_small_case():
    """Child process with TE_PINGPONG_MIN_MB=0: every eager dense tick is an A -> B tick (records read from one buffer,
    written with nontemporal stores to the other, buffers swapped).  Interleaved with everything that touches the CURRENT
    buffer in place -- one-target calls, by-id batches, erase, creation beyond the capacity (reallocation), predict-only
    ticks, the dense kernels of coupled matrices -- and compared with the oracle on every target."""
    from conftest import MODEL_FILES, model_path
    models = {k: oracle.load_model_yaml(model_path(k)) for k in MODEL_FILES}
    assert os.environ.get("TE_PINGPONG_MIN_MB") == "0"
    rng = np.random.default_rng(5)
    for dtype in ("f64", "f32"):
        for name, coupled in (("angular_rates", False), ("angular_velocities", False), ("uniform_acceleration", True), ("angular_rates", True)):
            m = models[name]
            Q, R, P0 = m["Q"], m["R"], m["P"]
            if coupled:   # coupled symmetric matrices -> the dense kernels (kf_step.hpp) take the A -> B path too
                def spd(A, s):
                    B = rng.normal(size=A.shape) * s
                    d = np.sqrt(np.diag(A))
                    return A + (B @ B.T) * np.outer(d, d)
                Q, R, P0 = spd(Q, 0.3), spd(R, 0.3), spd(P0, 0.3)
            N, ticks, dt = 333, 9, 0.004
            ref = oracle.stream_fill(m["model"], 77, N + 200, ticks, dt, dtype=dtype)
            from target_estimation_amd.streams import make_stream
            st = make_stream(m["model"], N + 200, ticks, dt, 77, dtype=dtype)
            ids = np.arange(N, dtype=np.uint32) + 10
            mgr = te.TargetManager(dtype=dtype)
            mgr.init_batch(ids, dt, 0.0, ref["p0"][:N], type=m["model"], Q=Q, R=R, P0=P0)
            b = mgr.batches()[0]
            orc = oracle.OracleBatch(m["model"], Q, R, P0, ref["p0"][:N], dt, dtype=dtype)
            meas = st["meas"]
            mgr.step_sequence_all(dt, [meas[0:3, :, :N].contiguous()], use_graph=0)        # 3 A -> B ticks (odd: ends in the other buffer)
            for s in range(3):
                orc.step(dt, ref["meas"][s, :N])
            check_state(mgr, ids, orc, dtype, "%s after the all-batches call" % name)
            b.step(dt, meas[3, :, :N].contiguous())                                        # one A -> B tick
            orc.step(dt, ref["meas"][3, :N])
            mgr.update(int(ids[5]), dt, ref["meas"][4, 5])                                  # one-target call: in place, current buffer
            mgr.update(int(ids[6]), dt)
            one = oracle.OracleBatch(m["model"], Q, R, P0, ref["p0"][:N], dt, dtype=dtype)   # oracle per target: replay target 5 / 6
            b.step_sequence(dt, meas[5:7, :, :N].contiguous(), None, use_graph=False)      # two more A -> B ticks
            b.step(dt, None)                                                               # predict only
            # the oracle, target by target, in the same order of operations
            mask5 = np.zeros(N, dtype=np.uint8); mask5[5] = 1
            mask6 = np.zeros(N, dtype=np.uint8); mask6[6] = 1
            del one
            _step_subset(orc, dt, ref["meas"][4, :N], mask5, predict_others=False)
            _step_subset(orc, dt, None, mask6, predict_others=False)
            for s in (5, 6):
                orc.step(dt, ref["meas"][s, :N])
            orc.step(dt, None)
            check_state(mgr, ids, orc, dtype, "%s after in-place calls between A -> B ticks" % name)
            # growth beyond the capacity reallocates the records (and drops the alternate buffer)
            more = np.arange(200, dtype=np.uint32) + 5000
            mgr.init_batch(more, dt, 0.0, ref["p0"][N:], type=m["model"], Q=Q, R=R, P0=P0)
            assert mgr.erase(int(ids[0]))
            b.step(dt, None)
            keep = np.concatenate([ids[1:], more])
            orc2 = oracle.OracleBatch(m["model"], Q, R, P0, ref["p0"][N:], dt, dtype=dtype)
            orc2.step(dt, None)
            orc.step(dt, None)
            x, P = mgr.get_state_batch(keep)
            xo, Po = orc.state(); xn, Pn = orc2.state()
            xo, Po = np.concatenate([xo[1:], xn]), np.concatenate([Po[1:], Pn])
            tol = 1e-9 if dtype == "f64" else 2e-3
            assert np.abs(x - xo).max() <= tol * (1 + np.abs(xo).max()), (name, dtype)
            assert (np.abs(P - Po).max(axis=(1, 2)) <= tol * np.abs(Po).max(axis=(1, 2))).all(), (name, dtype)
            mgr.close()
    print("pingpong small case ok")


def _step_subset(orc, dt, meas, mask, predict_others):
    """Step only the oracle targets whose mask byte is set (with `meas` or, if None, predict-only); the others stay."""
    import ctypes as C
    f_add = orc._f("orc_target_add_measurement")
    f_upd = orc._f("orc_target_update")
    for i in np.nonzero(mask)[0]:
        if meas is None:
            f_upd(orc._at(int(i)), float(dt))
        else:
            row = np.ascontiguousarray(meas[i], dtype=np.float64)
            f_add(orc._at(int(i)), float(dt), row.ctypes.data_as(C.POINTER(C.c_double)))


@pytest.mark.parametrize("case", ["pingpong"])
def test_ab_ticks_forced_on_small_batches_match_oracle(case):
    env = dict(os.environ, TE_PINGPONG_MIN_MB="0", PYTHONPATH=os.pathsep.join([os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__))]))
    p = subprocess.run([sys.executable, "-c", "import test_gpu_mixed_configs as t; t._pingpong_small_case()"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "pingpong small case ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_ab_ticks_equal_in_place_ticks_bit_for_bit():
    """The same 4 * 10^6-target batch stepped in place (policy off) and A -> B (policy on, the default at this size): the
    records must be the same bits.  Two child processes (the thresholds are read once per process), 20 000-target sample."""
    code = ("import numpy as np, torch, target_estimation_amd as te\n"
            "from target_estimation_amd.streams import make_stream\n"
            "import os, sys\n"
            "sys.path.insert(0, os.path.join(%r, 'tests'))\n"
            "from conftest import model_path\n"
            "N, T, dt = 4_000_000, 5, 0.004\n"
            "st = make_stream(1, N, T, dt, 11)\n"
            "mgr = te.TargetManager(model_path('angular_velocities'))\n"
            "ids = np.arange(N, dtype=np.uint32)\n"
            "mgr.init_batch(ids, dt, 0.0, st['p0'].cpu().numpy())\n"
            "b = mgr.batches()[0]\n"
            "b.step_sequence(dt, st['meas'], None, use_graph=False)\n"
            "b.step(dt, st['meas'][0])\n"
            "x, P = mgr.get_state_batch(ids[::200])\n"
            "np.save(sys.argv[1], np.concatenate([x.ravel(), P.ravel()]))\n" % os.path.dirname(os.path.dirname(__file__)))
    import tempfile
    out = []
    for mb in ("-1", "1024"):
        f = tempfile.NamedTemporaryFile(suffix=".npy", delete=False).name
        env = dict(os.environ, TE_PINGPONG_MIN_MB=mb)
        p = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out.append(np.load(f))
        os.unlink(f)
    assert np.isfinite(out[0]).all()
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_population_tick_of_four_models_equals_launches_per_batch(models, dtype):
    """Round 4: one launch steps every batch of a manager (kf_step_population_kernel).  All four motion models, ragged sizes
    (one of them less than a wavefront), a mask on one batch, eager and recorded, against a second manager stepped batch by
    batch with single launches: bit for bit; and the oracle on every target of the two smallest batches."""
    parts = [("angular_rates", 1003), ("angular_velocities", 517), ("uniform_acceleration", 40), ("uniform_velocity", 2050)]
    ticks, dt = 6, 0.004
    mgr, batches, out, meas = _build(models, parts, dtype, ticks, dt, 4100)
    ref, rbatches, _, rmeas = _build(models, parts, dtype, ticks, dt, 4100)
    assert mgr.population_tick() and ref.population_tick()
    rng = np.random.default_rng(3)
    has = [None, torch.as_tensor((rng.random((ticks, parts[1][1])) < 0.7).astype(np.uint8), device="cuda"), None, None]
    for s in range(ticks):
        for j, b in enumerate(rbatches):
            b.step(dt, rmeas[j][s], None if has[j] is None else has[j][s])
    mgr.step_sequence_all(dt, [m[:3] for m in meas], has_meas=[None if h is None else h[:3] for h in has], use_graph=0)
    mgr.step_sequence_all(dt, [m[3:] for m in meas], has_meas=[None if h is None else h[3:] for h in has], use_graph=1)
    torch.cuda.synchronize()
    for o in out:
        got, want = mgr.get_state_batch(o["ids"]), ref.get_state_batch(o["ids"])
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
    assert mgr.getNumberMeasurements(int(out[1]["ids"][5])) == int(has[1][:, 5].sum())
    for j in (2, 0):
        sample = np.arange(parts[j][1])
        check_state(mgr, out[j]["ids"], _oracle_on_sample(models, out[j], sample, ticks, dt, dtype), dtype, "%s in a population launch" % parts[j][0])
    ref.close(); mgr.close()


def test_population_tick_applies_only_where_every_batch_qualifies(models):
    """One launch per tick needs at least two non-empty batches, all of them one-class batches in the separable layout with
    packed groups; anything else keeps a launch per batch (and still gives the same results: the other tests of this file
    and test_gpu_intersection.py run the dense layouts through the same call)."""
    dt = 0.004
    one = np.tile([0, 0, 0, 0, 0, 0, 1.0], (70, 1))

    def manager(names, lanes=0, classes=False):
        mgr = te.TargetManager(dtype="f64", lanes_per_target=lanes)
        base = 0
        for name in names:
            m = models[name]
            ids = np.arange(70, dtype=np.uint32) + base
            base += 70
            mgr.init_batch(ids, dt, 0.0, one, type=te.MODEL_TYPES[name], Q=m["Q"], R=m["R"], P0=m["P"])
            if classes:   # a second (Q, R) class in the same batch
                ids2 = np.arange(30, dtype=np.uint32) + 100_000 + base
                mgr.init_batch(ids2, dt, 0.0, one[:30], type=te.MODEL_TYPES[name], Q=2.0 * m["Q"], R=m["R"], P0=m["P"])
        return mgr
    cases = [(manager(["angular_rates", "angular_velocities"]), True),
             (manager(["angular_rates"]), False),
             (manager(["angular_rates", "uniform_acceleration"], lanes=3), False),
             (manager(["angular_rates", "uniform_velocity"], classes=True), False)]
    for mgr, want in cases:
        assert mgr.population_tick() is want
        mgr.close()
    keep = manager(["uniform_velocity", "uniform_acceleration"])
    assert keep.population_tick()
    keep.set_keep_measurement(True)          # measured-pose rows are written by a kernel behind every batch's step
    assert not keep.population_tick()
    keep.close()
