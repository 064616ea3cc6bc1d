"""Resident ("live") mode (target_batch_live_*): ONE launch holds a small batch's state in registers and serves tick after tick
as the host posts them -- BASELINE.json configs[1] / configs[2] are launch-bound, a dependent launch costs more than their
tick.  Results must be those of single ticks bit for bit, and the oracle's within the usual tolerance.  What the session
replaces: the caller's per-tick loop over targets, src/target_manager.cpp:190-225 / src/target_node.cpp:36-44.

Every test drives the manager on its own NON-blocking stream and refills the ring on another one: work queued behind the
resident kernel on its stream (or on the legacy default stream) would wait for the session to end."""
import os

import numpy as np
import pytest

import oracle
from conftest import model_path
from test_gpu_parity import check_state

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")

MODELS = {"angular_rates": 0, "angular_velocities": 1, "uniform_acceleration": 2, "uniform_velocity": 3}


def _setup(models, name, dtype, N, ticks, dt, seed, availability=1.0):
    from target_estimation_amd.streams import make_stream
    live = torch.cuda.Stream()
    st = make_stream(MODELS[name], N, ticks, dt, seed, dtype=dtype, availability=availability)
    torch.cuda.synchronize()
    mgr = te.TargetManager(model_path(name), dtype=dtype)
    mgr.set_stream(live.cuda_stream)
    ids = np.arange(N, dtype=np.uint32) + 3
    p0 = st["p0"].cpu().numpy()
    mgr.init_batch(ids, dt, 0.0, p0)
    mgr.synchronize()
    return mgr, mgr.batches()[0], st, ids, p0, live


@pytest.mark.parametrize("name,dtype,N", [("uniform_velocity", "f64", 10_000), ("uniform_acceleration", "f32", 100_000),
                                          ("angular_rates", "f64", 3_000), ("angular_velocities", "f32", 5_001)])
def test_live_session_equals_single_ticks_bit_for_bit(models, name, dtype, N):
    """configs[1] / configs[2] sizes among them.  40 ticks through a ring of 16 that the host refills on a second stream while
    the session runs; the ticks posted one doorbell at a time, in bursts, and all at once; against the same ticks as single
    launches (bit for bit) and against the oracle on a sample."""
    ticks, ring, dt = 40, 16, 0.004
    m = models[name]
    mgr, b, st, ids, p0, live = _setup(models, name, dtype, N, ticks, dt, 91, availability=0.9)
    meas, has = st["meas"], st["has_meas"]
    # reference: single ticks on a second manager
    ref = te.TargetManager(model_path(name), dtype=dtype)
    ref.init_batch(ids, dt, 0.0, p0)
    rb = ref.batches()[0]
    for s in range(ticks):
        rb.step(dt, meas[s], has[s])
    want = ref.get_state_batch(ids)
    want_nm = [ref.getNumberMeasurements(int(i)) for i in ids[:50]]
    torch.cuda.synchronize()
    assert b.live_capacity >= N
    ring_m, ring_h = meas[:ring].clone(), has[:ring].clone()
    torch.cuda.synchronize()
    copy = torch.cuda.Stream()
    b.live_start(dt, ring_m, ring_h, max_ticks=ticks, idle_limit_s=3.0)
    b.live_post(1)                                   # one doorbell, one tick
    assert b.live_wait(1, 5.0) and b.live_done() >= 1
    b.live_post_each(ring - 1)                       # the rest of the ring, a doorbell per tick, back to back
    assert b.live_wait(ring, 5.0)
    done = ring
    while done < ticks:                              # refill the whole ring behind the session, post it as ONE doorbell
        k = min(ring, ticks - done)
        with torch.cuda.stream(copy):
            ring_m[:k].copy_(meas[done:done + k])
            ring_h[:k].copy_(has[done:done + k])
        copy.synchronize()
        b.live_post(k)
        done += k
        assert b.live_wait(done, 5.0)
    assert b.live_stop() == ticks
    got = mgr.get_state_batch(ids)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    assert [mgr.getNumberMeasurements(int(i)) for i in ids[:50]] == want_nm
    assert mgr.getTime(int(ids[0])) == pytest.approx(ticks * dt, abs=1e-12)
    sample = np.arange(0, N, max(1, N // 300))
    refs = oracle.stream_sample(m["model"], 91, sample, ticks, dt, availability=0.9, dtype=dtype)
    orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0[sample], dt, dtype=dtype)
    for s in range(ticks):
        orc.step(dt, refs["meas"][s], refs["has_meas"][s])
    check_state(mgr, ids[sample], orc, dtype, "%s after a live session" % name)
    ref.close(); mgr.close()


def test_any_other_call_ends_the_session_first(models):
    """While a session is open the records in HBM are stale: getters, steps, erase and init end it first (and see its ticks)."""
    name, dtype, N, dt = "uniform_acceleration", "f64", 2000, 0.004
    mgr, b, st, ids, p0, live = _setup(models, name, dtype, N, 12, dt, 5)
    meas = st["meas"]
    ref = te.TargetManager(model_path(name), dtype=dtype)
    ref.init_batch(ids, dt, 0.0, p0)
    rb = ref.batches()[0]
    b.live_start(dt, meas, max_ticks=100, idle_limit_s=3.0)
    b.live_post(5)
    pose = mgr.getTargetPose(int(ids[7]))[1]              # a getter: stops the session after its 5 ticks
    for s in range(5):
        rb.step(dt, meas[s])
    np.testing.assert_array_equal(pose, ref.getTargetPose(int(ids[7]))[1])
    assert b.live_done() == 0 and b.live_stop() == 0      # no session any more
    with pytest.raises(RuntimeError, match="without a live session"):
        b.live_post(1)
    b.live_start(dt, meas, first_entry=5, max_ticks=100, idle_limit_s=3.0)   # a second session continues with ring entry 5
    b.live_post(3)
    mgr.update(int(ids[0]), dt, p0[0])                    # queued one-target step ...
    assert mgr.erase(int(ids[1]))                         # ... and an erase: the session ends, then both run
    for s in range(5, 8):
        rb.step(dt, meas[s])
    ref.update(int(ids[0]), dt, p0[0])
    assert ref.erase(int(ids[1]))
    keep = np.delete(ids, 1)
    got, want = mgr.get_state_batch(keep), ref.get_state_batch(keep)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    ref.close(); mgr.close()


def test_live_mode_refuses_what_it_cannot_serve(models):
    name, dt = "uniform_velocity", 0.004
    m = models[name]
    live = torch.cuda.Stream()
    ring = torch.zeros((4, 7, 64), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    mgr = te.TargetManager(dtype="f64")
    mgr.set_stream(live.cuda_stream)
    rng = np.random.default_rng(0)
    B = rng.normal(size=(6, 6)) * 0.1
    Qc = m["Q"] + (B @ B.T) * 1e-6                        # coupled Q: the dense kernel's layout, no live kernel
    ids = np.arange(10, dtype=np.uint32)
    p0 = np.tile([0, 0, 0, 0, 0, 0, 1.0], (10, 1))
    mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=Qc, R=m["R"], P0=m["P"])
    b = mgr.batches()[0]
    assert b.live_capacity == 0
    with pytest.raises(RuntimeError, match="axis-separable"):
        b.live_start(dt, ring)
    mgr.close()
    mgr = te.TargetManager(model_path(name))
    mgr.set_stream(live.cuda_stream)
    mgr.init_batch(ids, dt, 0.0, p0)
    mgr.init_batch(ids + 100, dt, 0.0, p0, type=m["model"], Q=m["Q"] * 2, R=m["R"], P0=m["P"])   # a second (Q, R) class
    b = mgr.batches()[0]
    assert b.num_classes == 2
    with pytest.raises(RuntimeError, match="one \\(Q, R\\) class"):
        b.live_start(dt, ring)
    mgr.close()
    mgr = te.TargetManager(model_path(name))
    mgr.set_stream(live.cuda_stream)
    mgr.init_batch(ids, dt, 0.0, p0)
    b = mgr.batches()[0]
    with pytest.raises(RuntimeError, match="bad measurement ring"):
        b._lib.target_batch_live_start(b._h, dt, ring.data_ptr(), 7 * 64, 4, None, 0, 4, 0, 10, 1.0) and None
        raise RuntimeError(te.capi.last_error())
    b.live_start(dt, ring, max_ticks=3, idle_limit_s=3.0)
    b.live_post(3)
    with pytest.raises(RuntimeError, match="beyond the session"):
        b.live_post(1)
    assert b.live_wait(3, 5.0) and b.live_stop() == 3
    mgr.close()


def test_an_abandoned_session_ends_by_itself(models):
    """A host that stops posting (or dies) must not leave a kernel behind: after idle_limit_s without news every wavefront
    stores its records and exits; stop() then finds an even, complete session."""
    import time
    mgr, b, st, ids, p0, live = _setup(models, "uniform_velocity", "f64", 5000, 4, 0.004, 8)
    b.live_start(0.004, st["meas"], max_ticks=100, idle_limit_s=0.5)
    b.live_post(2)
    assert b.live_wait(2, 5.0)
    t0 = time.time()
    assert b.live_running()
    while b.live_running() and time.time() - t0 < 8.0:    # the relay's last store says that it has left
        time.sleep(0.001)
    waited = time.time() - t0
    assert 0.4 < waited < 1.0, waited                     # the limit is a TIME on the device's clock (0.5 s asked for)
    assert b.live_stop() == 2
    x, P = mgr.get_state_batch(ids[:10])
    assert np.isfinite(x).all() and np.isfinite(P).all()
    mgr.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_two_models_resident_together_on_the_default_stream(models, dtype):
    """BASELINE configs[3] at its per-GPU share (62 500 angular-rates + 62 500 angular-velocities): both batches of ONE manager
    in live mode at the same time (each resident kernel on its own stream), the manager itself on the default stream, other
    work of the manager going on meanwhile; against the all-batches call with a launch per tick, bit for bit, and the oracle."""
    import bench
    from target_estimation_amd.streams import make_stream
    parts = bench.MIXED["cfg4"][1]
    if dtype == "f64":
        # Round 3: the fp64 angular kernels held a wavefront's state in 262 / 282 registers, one wavefront per SIMD, and the
        # configs[3] share did not fit.  Now (part of the record parked in LDS, no loop-invariant constants in vector registers:
        # 125-167 registers, three wavefronts per SIMD) it does and runs below at the full share.  What does NOT fit any more is
        # 200 000 + 200 000: the manager must say so instead of starting a session that cannot be fully resident.
        mgr = te.TargetManager(dtype=dtype)
        rings, base = [], 0
        for k, (name, n) in enumerate([(name, 200_000) for name, _ in parts]):
            m = models[name]
            ids = np.arange(n, dtype=np.uint32) + base
            base += n
            mgr.init_batch(ids, 0.004, 0.0, np.tile([0, 0, 0, 0, 0, 0, 1.0], (n, 1)), type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
            rings.append(torch.zeros((2, 7, n), dtype=torch.float64, device="cuda"))
        with pytest.raises(RuntimeError, match="do not fit the device together"):
            mgr.live_start_all(0.004, rings)
        assert not any(b._lib.target_batch_live_done(b._h) for b in mgr.batches())
        mgr.close()
    ticks, dt = 12, 0.004

    def build():
        mgr = te.TargetManager(dtype=dtype)
        base, meas, info = 0, [], []
        for k, (name, n) in enumerate(parts):
            m = models[name]
            st = make_stream(MODELS[name], n, ticks, dt, 300 + k, dtype=dtype)
            ids = np.arange(n, dtype=np.uint32) + base
            base += n
            p0 = st["p0"].cpu().numpy()
            mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
            meas.append(st["meas"])
            info.append((name, ids, p0, 300 + k))
        return mgr, meas, info
    ref, rmeas, _ = build()
    ref.step_sequence_all(dt, rmeas, use_graph=0)
    mgr, meas, info = build()
    torch.cuda.synchronize()
    mgr.live_start_all(dt, meas, max_ticks=ticks, idle_limit_s=3.0)
    mgr.live_post_all(5, one_doorbell_per_tick=True)
    assert mgr.live_wait_all(5, 5.0) and mgr.live_done_all() >= 5
    assert mgr.size() == sum(n for _, n in parts)                  # a call that does not touch the batches leaves the sessions alone
    assert all(b.live_done() >= 5 for b in mgr.batches())
    mgr.live_post_all(ticks - 5)
    assert mgr.live_wait_all(ticks, 5.0)
    assert mgr.live_stop_all() == ticks
    for (name, ids, p0, seed), b in zip(info, mgr.batches()):
        got, want = mgr.get_state_batch(ids[::37]), ref.get_state_batch(ids[::37])
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        m = models[name]
        sample = np.arange(0, len(ids), 400)
        rs = oracle.stream_sample(m["model"], seed, sample, ticks, dt, dtype=dtype)
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0[sample], dt, dtype=dtype)
        for s in range(ticks):
            orc.step(dt, rs["meas"][s])
        check_state(mgr, ids[sample], orc, dtype, "%s, two models resident together" % name)
    ref.close(); mgr.close()


def test_config4_share_resident_with_the_per_tick_query(models):
    """BASELINE configs[4] at its per-GPU share (62 500 angular-rates + 62 500 uniform-acceleration, fp32, the sphere query of
    every target after every tick): both batches resident, the query inside the resident kernels, against the all-batches
    call with a launch per tick (state, delta and pose after the last tick: bit for bit) and the oracle."""
    import bench
    from target_estimation_amd.streams import make_stream
    parts, dtype = bench.MIXED["cfg5"][1], "f32"
    ticks, dt = 9, 0.004
    origin, radius = np.zeros(3), 5.0

    def build():
        mgr = te.TargetManager(dtype=dtype)
        base, meas, info = 0, [], []
        for k, (name, n) in enumerate(parts):
            m = models[name]
            st = make_stream(MODELS[name], n, ticks, dt, 700 + k, dtype=dtype)
            ids = np.arange(n, dtype=np.uint32) + base
            base += n
            p0 = st["p0"].cpu().numpy()
            rng = np.random.default_rng(40 + k)             # inbound, accelerating targets: the query has roots to find
            d = p0[:, :3] / np.linalg.norm(p0[:, :3], axis=1, keepdims=True)
            v0 = np.concatenate([-d * rng.uniform(1, 6, (n, 1)), np.zeros((n, 3))], 1)
            a0 = np.concatenate([rng.normal(0, 1.0, (n, 3)) + [0, 0, -2.0], np.zeros((n, 3))], 1)
            mgr.init_batch(ids, dt, 0.0, p0, v0, a0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
            meas.append(st["meas"])
            info.append((name, ids, p0, v0, a0, 700 + k))
        deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in mgr.batches()]
        poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in mgr.batches()]
        return mgr, meas, info, deltas, poses
    ref, rmeas, _, rd, rp = build()
    ref.step_sequence_all(dt, rmeas, query=(origin, radius, rd, rp), use_graph=0)
    mgr, meas, info, dd, pp = build()
    torch.cuda.synchronize()
    mgr.live_start_all(dt, meas, max_ticks=ticks, idle_limit_s=3.0, query=(origin, radius, dd, pp))
    mgr.live_post_all(ticks, one_doorbell_per_tick=True)
    assert mgr.live_wait_all(ticks, 5.0) and mgr.live_stop_all() == ticks
    torch.cuda.synchronize()
    hits = 0
    for j, (name, ids, p0, v0, a0, seed) in enumerate(info):
        np.testing.assert_array_equal(dd[j].cpu().numpy(), rd[j].cpu().numpy())
        np.testing.assert_array_equal(pp[j].cpu().numpy(), rp[j].cpu().numpy())
        got, want = mgr.get_state_batch(ids[::53]), ref.get_state_batch(ids[::53])
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        m = models[name]
        sample = np.arange(0, len(ids), 250)
        rs = oracle.stream_sample(m["model"], seed, sample, ticks, dt, dtype=dtype)
        orc = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0[sample], dt, 0.0, v0[sample], a0[sample], dtype=dtype)
        for s in range(ticks):
            orc.step(dt, rs["meas"][s])
        ok_o, pose_o, delta_o = orc.intersection_pose(ticks * dt, origin, radius)
        d = dd[j].cpu().numpy()[sample]
        both = (d > -1) & (delta_o > -1)
        assert ((d > -1) != (delta_o > -1)).mean() <= 0.01
        np.testing.assert_allclose(d[both], delta_o[both], rtol=5e-4, atol=5e-4)
        hits += int(both.sum())
    assert hits > 50
    ref.close(); mgr.close()


def test_query_results_of_a_live_session_are_readable_tick_by_tick(models):
    """The per-tick sphere query of a resident session writes delta / pose "overwritten every tick": a consumer on another
    stream that copies them once the tick's completion has reached the host must get THAT tick's results (they leave the
    kernel with write-through stores ahead of the progress word), not what an XCD's L2 still holds -- against a manager stepped
    by single launches with the fused query, bit for bit, tick by tick, without ending the session."""
    from target_estimation_amd.streams import make_stream
    parts, dtype = [("angular_rates", 6000), ("uniform_acceleration", 5000)], "f32"
    ticks, dt = 8, 0.004
    origin, radius = np.zeros(3), 5.0

    def build():
        mgr = te.TargetManager(dtype=dtype)
        base, meas = 0, []
        for k, (name, n) in enumerate(parts):
            m = models[name]
            st = make_stream(MODELS[name], n, ticks, dt, 900 + k, dtype=dtype)
            ids = np.arange(n, dtype=np.uint32) + base
            base += n
            p0 = st["p0"].cpu().numpy()
            rng = np.random.default_rng(60 + k)
            d = p0[:, :3] / np.linalg.norm(p0[:, :3], axis=1, keepdims=True)
            v0 = np.concatenate([-d * rng.uniform(1, 6, (n, 1)), np.zeros((n, 3))], 1)
            a0 = np.concatenate([rng.normal(0, 1.0, (n, 3)) + [0, 0, -2.0], np.zeros((n, 3))], 1)
            mgr.init_batch(ids, dt, 0.0, p0, v0, a0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
            meas.append(st["meas"])
        deltas = [torch.full((b.size,), 123.0, dtype=torch.float64, device="cuda") for b in mgr.batches()]
        poses = [torch.zeros((b.size, 7), dtype=torch.float64, device="cuda") for b in mgr.batches()]
        return mgr, meas, deltas, poses
    ref, rmeas, rd, rp = build()
    want = []
    for s in range(ticks):
        ref.step_sequence_all(dt, [m[s:s + 1] for m in rmeas], query=(origin, radius, rd, rp), use_graph=0)
        torch.cuda.synchronize()
        want.append(([d.cpu().numpy().copy() for d in rd], [q.cpu().numpy().copy() for q in rp]))
    mgr, meas, dd, pp = build()
    hd = [torch.empty_like(d, device="cpu").pin_memory() for d in dd]
    hp = [torch.empty_like(q, device="cpu").pin_memory() for q in pp]
    torch.cuda.synchronize()
    copy = torch.cuda.Stream()
    mgr.live_start_all(dt, meas, max_ticks=ticks, idle_limit_s=3.0, query=(origin, radius, dd, pp))
    changed = 0
    for s in range(ticks):
        mgr.live_post_all(1)
        assert mgr.live_wait_all(s + 1, 5.0)
        with torch.cuda.stream(copy):
            for j in range(len(dd)):
                hd[j].copy_(dd[j], non_blocking=True)
                hp[j].copy_(pp[j], non_blocking=True)
        copy.synchronize()
        for j in range(len(dd)):
            np.testing.assert_array_equal(hd[j].numpy(), want[s][0][j])
            np.testing.assert_array_equal(hp[j].numpy(), want[s][1][j])
            if s:
                changed += int((want[s][0][j] != want[s - 1][0][j]).sum())
    assert changed > 1000                                   # the results do move from tick to tick: a stale copy would be caught
    assert mgr.live_stop_all() == ticks
    ref.close(); mgr.close()


def test_a_busy_session_is_not_an_idle_one(models):
    """The idle limit counts rounds in which NOTHING happens.  A burst that takes longer to serve than the limit (the host is
    silent because it waits for it) must not end the session: later doorbells are still served."""
    import time
    mgr, b, st, ids, p0, live = _setup(models, "uniform_acceleration", "f32", 100_000, 8, 0.004, 3)
    limit, burst = 0.05, 100_000                                                # 50 ms; ~120 ms of work in one doorbell
    for attempt in range(3):
        b.live_start(0.004, st["meas"], max_ticks=1 << 20, idle_limit_s=limit)
        t0 = time.perf_counter()
        b.live_post(burst)
        assert b.live_wait(burst, 10.0)
        busy = time.perf_counter() - t0
        t1 = time.perf_counter()
        b.live_post(7)
        gap = time.perf_counter() - t1                                          # what THIS process took between the completion and its next doorbell
        served = b.live_wait(burst + 7, 5.0)
        if served:
            assert busy > 1.5 * limit, busy                                     # the burst really outlasted the idle limit
            assert b.live_stop() == burst + 7
            break
        # the session ended before the doorbell: legitimate only if the HOST was late (a descheduled process on a busy box), and then
        # the session must have ended cleanly at the burst
        assert gap > 0.25 * limit or not b.live_running(), (attempt, gap)
        with pytest.raises(RuntimeError, match="ended after"):
            b.live_stop()
    else:
        pytest.fail("three attempts, the host late every time")
    x, P = mgr.get_state_batch(ids[:16])
    assert np.isfinite(x).all() and np.isfinite(P).all()
    mgr.close()


@pytest.mark.parametrize("name,dtype,N", [("uniform_velocity", "f64", 10_000), ("angular_velocities", "f32", 30_000)])
def test_per_tick_pose_output_of_a_live_session(models, name, dtype, N):
    """The node loop of the reference publishes every target's filtered pose every tick (src/target_manager_ros.cpp:78-87).  A
    live session writes them to a device buffer after every tick: posted one tick at a time, the buffer copied on a second stream
    after each tick's completion holds exactly that tick's poses -- the getter's poses of a manager stepped by single launches,
    bit for bit -- without ending the session."""
    ticks, dt = 10, 0.004
    mgr, b, st, ids, p0, live = _setup(models, name, dtype, N, ticks, dt, 61)
    ref = te.TargetManager(model_path(name), dtype=dtype)
    ref.init_batch(ids, dt, 0.0, p0)
    rb = ref.batches()[0]
    want = []
    for s in range(ticks):
        rb.step(dt, st["meas"][s])
        want.append(rb.get_est(twist=False, acc=False)[0].cpu().numpy())       # [N, 7]
    torch.cuda.synchronize()
    ld = N + 13                                                                # a padded row length
    pose_soa = torch.full((7, ld), float("nan"), dtype=torch.float64, device="cuda")
    host = torch.empty((7, ld), dtype=torch.float64).pin_memory()
    torch.cuda.synchronize()
    copy = torch.cuda.Stream()
    b.live_set_pose_output(pose_soa)
    b.live_start(dt, st["meas"], max_ticks=ticks, idle_limit_s=3.0)
    for s in range(ticks):
        b.live_post(1)
        assert b.live_wait(s + 1, 5.0)
        with torch.cuda.stream(copy):
            host.copy_(pose_soa, non_blocking=True)
        copy.synchronize()
        np.testing.assert_array_equal(host.numpy()[:, :N].T, want[s])
        assert np.isnan(host.numpy()[:, N:]).all()                             # nothing beyond the batch is written
    assert b.live_done() == ticks and b.live_stop() == ticks
    b.live_set_pose_output(None)
    ref.close(); mgr.close()


def test_ring_refill_on_another_stream_next_to_a_nearly_full_device(models):
    """A session at 97 % of the capacity the library allows (uniform acceleration fp32: 18 of at most 19 resident wavefronts
    per CU) must leave other streams alive: the device-to-device copy that refills the ring completes in its usual time and the
    session goes on.  (Before the capacity was limited to 5 wavefronts per SIMD such a copy waited for the session to end.)"""
    import time
    name, dtype, dt = "uniform_acceleration", "f32", 0.004
    probe = te.TargetManager(model_path(name), dtype=dtype)
    p1 = np.zeros((64, 7)); p1[:, 6] = 1
    probe.init_batch(np.arange(64, dtype=np.uint32), dt, 0.0, p1)
    cap = probe.batches()[0].live_capacity
    probe.close()
    N = int(cap * 0.97) // 64 * 64
    mgr, b, st, ids, p0, live = _setup(models, name, dtype, N, 32, dt, 17)
    meas = st["meas"]
    ring = meas[:16].clone()
    torch.cuda.synchronize()
    copy = torch.cuda.Stream()
    b.live_start(dt, ring, max_ticks=32, idle_limit_s=3.0)
    b.live_post(16)
    assert b.live_wait(16, 5.0)
    t0 = time.perf_counter()
    with torch.cuda.stream(copy):
        ring[:16].copy_(meas[16:32])
    copy.synchronize()
    assert time.perf_counter() - t0 < 0.5, "the refill waited for the resident kernel"
    b.live_post(16)
    assert b.live_wait(32, 5.0)
    assert b.live_stop() == 32
    with pytest.raises(RuntimeError, match="resident wavefronts"):                 # beyond the capacity: refused, not attempted
        big = te.TargetManager(model_path(name), dtype=dtype)
        q = np.zeros((cap + 64, 7)); q[:, 6] = 1
        big.init_batch(np.arange(cap + 64, dtype=np.uint32), dt, 0.0, q)
        try:
            big.batches()[0].live_start(dt, torch.zeros((1, 7, cap + 64), dtype=torch.float32, device="cuda"), max_ticks=1)
        finally:
            big.close()
    mgr.close()


def _one_hardware_queue_case():
    """Child process with GPU_MAX_HW_QUEUES=1: every stream of one priority shares ONE hardware queue.  (1) a single session
    still lets a copy on another stream through (the resident kernel's stream has the high-priority class to itself);
    (2) a two-batch session either runs (the runtime gave each high-priority stream its own queue) or, with both resident
    kernels in one queue, is refused after the start timeout with a message; nothing hangs, the manager works afterwards."""
    import os
    import time
    from conftest import MODEL_FILES
    from target_estimation_amd.streams import make_stream
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "1"
    models = {k: oracle.load_model_yaml(model_path(k)) for k in MODEL_FILES}
    name, dtype, N, dt = "uniform_velocity", "f64", 10_000, 0.004
    mgr, b, st, ids, p0, live = _setup(models, name, dtype, N, 32, dt, 5)
    ring = st["meas"][:16].clone()
    torch.cuda.synchronize()
    copy = torch.cuda.Stream()
    b.live_start(dt, ring, max_ticks=32, idle_limit_s=3.0)
    b.live_post(16)
    assert b.live_wait(16, 5.0)
    t0 = time.perf_counter()
    with torch.cuda.stream(copy):
        ring[:16].copy_(st["meas"][16:32])
    copy.synchronize()
    assert time.perf_counter() - t0 < 0.5
    b.live_post(16)
    assert b.live_wait(32, 5.0) and b.live_stop() == 32
    mgr.close()
    # two batches
    mgr = te.TargetManager(dtype=dtype)
    mgr.set_stream(torch.cuda.Stream().cuda_stream)
    rings = []
    for k, nm in enumerate(("uniform_velocity", "uniform_acceleration")):
        m = models[nm]
        s2 = make_stream(MODELS[nm], 2000, 4, dt, 9 + k, dtype=dtype)
        mgr.init_batch(np.arange(2000, dtype=np.uint32) + 10_000 * k, dt, 0.0, s2["p0"].cpu().numpy(), type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
        rings.append(s2["meas"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        mgr.live_start_all(dt, rings, max_ticks=4, idle_limit_s=1.0)
        started = True
    except RuntimeError as e:
        started = False
        assert "did not start within" in str(e) or "do not run side by side" in str(e), str(e)
    assert time.perf_counter() - t0 < 30.0
    if started:      # (this runtime gives every high-priority stream a queue of its own: then the session simply works)
        mgr.live_post_all(4)
        assert mgr.live_wait_all(4, 5.0) and mgr.live_stop_all() == 4
        print("two resident kernels ran with GPU_MAX_HW_QUEUES=1")
    else:
        print("two resident kernels in one queue: refused")
    x, P = mgr.get_state_batch(np.arange(5, dtype=np.uint32))                      # the manager is usable afterwards
    assert np.isfinite(x).all()
    for bb, r in zip(mgr.batches(), rings):                                        # and ordinary ticks still run
        bb.step(dt, r[0])
    torch.cuda.synchronize()
    mgr.close()
    print("one hardware queue ok")


def test_sessions_with_one_hardware_queue_per_priority():
    import os
    import subprocess
    import sys
    # (the child also keeps the doorbell in HOST memory -- TE_LIVE_DOORBELL=host, the form for systems without a large BAR -- so that
    # both placements stay covered: every other test of this file runs with the doorbell behind the BAR where the box has one)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1", TE_LIVE_DOORBELL="host",
               PYTHONPATH=os.pathsep.join([os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__))]))
    p = subprocess.run([sys.executable, "-c", "import test_gpu_live as t; t._one_hardware_queue_case()"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "one hardware queue ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_a_second_session_that_does_not_fit_next_to_the_first_is_refused_cleanly(models):
    """Two managers, each within ITS capacity, together more than the device holds.  The library counts the resident sessions
    of the process, so the second start is refused at once.  With that count switched off (TE_LIVE_IGNORE_OTHERS=1: what a
    session of ANOTHER process looks like) the second grid is only partly resident and its last workgroup (the relay) never
    starts: live_start must say so after its start timeout, get the workers that did start out of the way (the stop word goes
    into their mirror words from the host), leave the second manager usable at once -- and the first session must go on
    serving ticks.  Calls that free device memory next to the resident session return at once (the frees wait on a list)."""
    import time
    name, dtype, dt = "uniform_acceleration", "f32", 0.004
    probe = te.TargetManager(model_path(name), dtype=dtype)
    p1 = np.zeros((64, 7)); p1[:, 6] = 1
    probe.init_batch(np.arange(64, dtype=np.uint32), dt, 0.0, p1)
    cap = probe.batches()[0].live_capacity
    probe.close()
    na, nb = int(cap * 0.95) // 64 * 64, int(cap * 0.6) // 64 * 64
    mgr_a, a, st_a, ids_a, p0_a, live_a = _setup(models, name, dtype, na, 8, dt, 3)
    mgr_b, b, st_b, ids_b, p0_b, live_b = _setup(models, name, dtype, nb, 8, dt, 4)
    a.live_start(dt, st_a["meas"], max_ticks=8, idle_limit_s=20.0)
    a.live_post(2)
    assert a.live_wait(2, 5.0)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="do not fit the device together"):
        b.live_start(dt, st_b["meas"], max_ticks=8, idle_limit_s=20.0)
    assert time.perf_counter() - t0 < 0.5
    os.environ["TE_LIVE_IGNORE_OTHERS"] = "1"
    try:
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="did not start within"):
            b.live_start(dt, st_b["meas"], max_ticks=8, idle_limit_s=20.0)
        assert time.perf_counter() - t0 < 6.0
    finally:
        del os.environ["TE_LIVE_IGNORE_OTHERS"]
    t0 = time.perf_counter()
    ok, pose = mgr_b.getTargetPose(int(ids_b[5]))               # the refused manager: usable at once, records untouched
    assert ok and np.allclose(pose[:3], p0_b[5, :3])
    b.step(dt, st_b["meas"][0])
    mgr_b.synchronize()
    # calls that FREE device memory (hipFree synchronises the device: next to the resident session it would block until the
    # session's idle limit, 20 s here): the measured-pose rows of a batch, and a batch that grows
    mgr_b.set_keep_measurement(True)
    mgr_b.set_keep_measurement(False)
    grow = np.arange(nb, nb + 70_000, dtype=np.uint32) + 10_000_000
    pg = np.zeros((len(grow), 7)); pg[:, 6] = 1
    assert mgr_b.init_batch(grow, dt, 0.0, pg) == len(grow)
    assert mgr_b.erase_batch(grow) == len(grow)
    mgr_b.synchronize()
    assert time.perf_counter() - t0 < 3.0
    a.live_post(6)                                              # the first session never noticed
    assert a.live_wait(8, 5.0) and a.live_stop() == 8
    ref = te.TargetManager(model_path(name), dtype=dtype)
    ref.init_batch(ids_b, dt, 0.0, p0_b)
    ref.batches()[0].step(dt, st_b["meas"][0])
    want, got = ref.get_state_batch(ids_b[::997]), mgr_b.get_state_batch(ids_b[::997])
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    ref.close(); mgr_a.close(); mgr_b.close()
