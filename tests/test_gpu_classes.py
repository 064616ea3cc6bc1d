"""Per-target model parameters (TargetManager::init takes Q, R, P0 per target, target_manager.hpp:85-87) without a batch
per parameter set: all (Q, R) classes of one layout live in ONE batch (a table in HBM + a class index per slot), so a
tick is one launch per motion model however many classes there are.  GPU vs the oracle, which is built per class."""
import numpy as np
import pytest

import oracle
from conftest import HARNESS_ORDER, synth_stream
from test_gpu_parity import TOL, to_soa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def _scaled_classes(m, n_classes, rng):
    s = rng.uniform(0.5, 2.0, (n_classes, 3))
    return (m["Q"][None] * s[:, 0, None, None], m["R"][None] * s[:, 1, None, None], m["P"][None] * s[:, 2, None, None])


def _coupled_classes(m, n_classes, rng):
    def spd(A, k):
        B = rng.normal(size=(k,) + A.shape) * 0.3
        d = np.sqrt(np.diag(A))
        return A[None] + (B @ B.transpose(0, 2, 1)) * np.outer(d, d)[None]
    return spd(m["Q"], n_classes), spd(m["R"], n_classes), spd(m["P"], n_classes)


def _check_classes(mgr, orcs, ids, members, dtype, what):
    t = TOL[dtype]
    worst = 0.0
    for c, orc in orcs.items():
        x, P = mgr.get_state_batch(ids[members[c]])
        xo, Po = orc.state()
        scale = np.abs(Po).max(axis=(1, 2), keepdims=True)
        ex = np.abs(x - xo) - (t["x_atol"] + t["x_rtol"] * np.abs(xo))
        assert ex.max() <= 0, "%s class %d: x error %.3e over tolerance" % (what, c, ex.max())
        eP = (np.abs(P - Po) / scale).max()
        assert eP <= t["P_rel"], "%s class %d: P error %.3e" % (what, c, eP)
        worst = max(worst, eP)
    return worst


@pytest.mark.parametrize("order", ["random", "runs"])
@pytest.mark.parametrize("name,dtype", [("angular_rates", "f64"), ("uniform_velocity", "f64"), ("angular_velocities", "f32"),
                                         ("uniform_acceleration", "f32")])
def test_hundred_thousand_targets_thousand_classes_one_batch(models, name, dtype, order):
    """10^5 targets, 10^3 distinct (Q, R, P0): one batch (= one launch per tick), parity vs the oracle on 40 classes.
    order = "random": every target draws its class, so every wavefront mixes classes (per-lane gathers of the class rows);
    "runs": targets of a class are neighbours (runs of about 100), so most wavefronts take the wave-uniform path that reads the
    row through the scalar cache, and the wavefronts that straddle two runs the other one."""
    m = models[name]
    N, NC, dt, ticks = 100_000, 1000, 0.004, 5
    rng = np.random.default_rng(5)
    Q, R, P0 = _scaled_classes(m, NC, rng)
    class_of = rng.integers(0, NC, N).astype(np.uint32)
    if order == "runs":
        class_of = np.sort(class_of)
    from target_estimation_amd.streams import make_stream
    st = make_stream(te.MODEL_TYPES[name], N, ticks, dt, 99)
    p0 = st["p0"].cpu().numpy()
    ids = (np.arange(N, dtype=np.uint32) * 3 + 1)
    mgr = te.TargetManager(dtype=dtype)
    assert mgr.init_batch_classes(ids, dt, 0.0, p0, te.MODEL_TYPES[name], Q, R, P0, class_of) == N
    batches = mgr.batches()
    assert len(batches) == 1 and batches[0].size == N and batches[0].num_classes == NC
    b = batches[0]
    assert b.layout == "axis_separable_packed"      # scaled copies of the shipped matrices stay separable and symmetric
    meas = st["meas"].to(b.torch_dtype()).contiguous()
    mask = (torch.rand(N, device="cuda") < 0.8).to(torch.uint8)
    for s in range(ticks):
        b.step(dt, meas[s], mask if s == 2 else None)
    chosen = rng.choice(NC, 40, replace=False)
    members = {int(c): np.nonzero(class_of == c)[0] for c in chosen}
    mh = meas.to(torch.float64).cpu().numpy()
    mask_h = mask.cpu().numpy()
    orcs = {}
    for c, rows in members.items():
        orc = oracle.OracleBatch(m["model"], Q[c], R[c], P0[c], p0[rows], dt, dtype=dtype)
        for s in range(ticks):
            orc.step(dt, np.ascontiguousarray(mh[s][:, rows].T), mask_h[rows] if s == 2 else None)
        orcs[c] = orc
    _check_classes(mgr, orcs, ids, members, dtype, name)
    # the classes really differ: class c's covariance is not class c2's
    c1, c2 = int(chosen[0]), int(chosen[1])
    _, P1 = mgr.get_state_batch(ids[members[c1]][:1])
    _, P2 = mgr.get_state_batch(ids[members[c2]][:1])
    assert np.abs(P1 - P2).max() > 1e-3 * np.abs(P1).max()
    mgr.close()


@pytest.mark.parametrize("name", HARNESS_ORDER)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_coupled_classes_indexed_scalar_and_erase(models, name, dtype):
    """Coupled (general) matrices -> the dense per-class kernel; the by-id and one-target paths and erase (which moves
    records, and their class index with them) on a multi-class batch.  One oracle target per GPU target."""
    m = models[name]
    N, NC, dt, steps = 600, 12, 0.004, 6
    rng = np.random.default_rng(11)
    Q, R, P0 = _coupled_classes(m, NC, rng)
    class_of = rng.integers(0, NC, N).astype(np.uint32)
    p0, meas = synth_stream(name, N, steps, seed=3)
    ids = rng.permutation(5000)[:N].astype(np.uint32)
    mgr = te.TargetManager(dtype=dtype)
    assert mgr.init_batch_classes(ids, dt, 0.0, p0, m["model"], Q, R, P0, class_of) == N
    assert len(mgr.batches()) == 1 and mgr.batches()[0].num_classes == NC and mgr.batches()[0].layout == "symmetric_packed"
    b = mgr.batches()[0]
    orcs = [oracle.OracleTarget(m["model"], Q[c], R[c], P0[c], p0[i], dt, dtype=dtype) for i, c in enumerate(class_of)]

    def compare(rows, what):
        t = TOL[dtype]
        x, P = mgr.get_state_batch(ids[rows])
        xo = np.concatenate([orcs[i].state()[0] for i in rows]); Po = np.concatenate([orcs[i].state()[1] for i in rows])
        assert (np.abs(x - xo) - (t["x_atol"] + t["x_rtol"] * np.abs(xo))).max() <= 0, what
        assert (np.abs(P - Po) / np.abs(Po).max(axis=(1, 2), keepdims=True)).max() <= t["P_rel"], what

    allrows = np.arange(N)
    for s in range(steps):
        if s % 3 == 0:      # dense device tick
            b.step(dt, to_soa(meas[s], b))
            for i in allrows:
                orcs[i].add_measurement(dt, meas[s][i])
        elif s % 3 == 1:    # by-id batch, random order, subset, mask
            sub = rng.permutation(N)[: N * 2 // 3]
            has = (rng.random(len(sub)) < 0.7).astype(np.uint8)
            assert mgr.update_batch(ids[sub], dt, meas[s][sub], has) == len(sub)
            for j, i in enumerate(sub):
                if has[j]:
                    orcs[i].add_measurement(dt, meas[s][i])
                else:
                    orcs[i].update(dt)
        else:               # queued one-target calls
            sub = rng.choice(N, 15, replace=False)
            for i in sub:
                mgr.update(int(ids[i]), dt, meas[s][i])
                orcs[i].add_measurement(dt, meas[s][i])
    compare(allrows, "%s after the mixed schedule" % name)
    # erase a third: the survivors (moved records) keep their class
    gone = rng.choice(N, N // 3, replace=False)
    assert mgr.erase_batch(ids[gone]) == len(gone)
    keep = np.setdiff1d(allrows, gone)
    b.step(dt, to_soa(meas[0][keep_order(b, ids)], b))   # a tick on the compacted batch, rows in the new slot order
    for i in keep:
        orcs[i].add_measurement(dt, meas[0][i])
    compare(keep, "%s after erase" % name)
    mgr.close()


def keep_order(batch, ids):
    """row (position in `ids`) of every slot of the batch, in slot order"""
    pos = {int(v): k for k, v in enumerate(ids)}
    return np.array([pos[int(v)] for v in batch.slot_ids()], dtype=np.int64)


def test_one_at_a_time_inits_join_one_batch(models):
    """The reference-order initialiser with a different (Q, R) per call: same model + same layout -> the same batch,
    one new class per distinct pair; a repeated pair reuses its class."""
    name = "uniform_velocity"
    m = models[name]
    mgr = te.TargetManager(dtype="f64")
    p0 = np.array([0, 0, 0, 0, 0, 0, 1.0])
    for k in range(6):
        s = 1.0 + (k % 3)
        mgr.init(k, 0.004, 0.0, p0, type=m["model"], Q=m["Q"] * s, R=m["R"] * s, P0=m["P"])
    assert len(mgr.batches()) == 1 and mgr.batches()[0].num_classes == 3 and mgr.batches()[0].size == 6
    meas = p0.copy(); meas[:3] = 0.01
    for k in range(6):
        mgr.update(k, 0.004, meas)
    x, P = mgr.get_state_batch(np.arange(6, dtype=np.uint32))
    for k in range(6):
        s = 1.0 + (k % 3)
        orc = oracle.OracleTarget(m["model"], m["Q"] * s, m["R"] * s, m["P"], p0, 0.004)
        orc.add_measurement(0.004, meas)
        xo, Po = orc.state()
        np.testing.assert_allclose(x[k], xo[0], atol=1e-12)
        np.testing.assert_allclose(P[k], Po[0], atol=1e-12 * np.abs(Po).max())
    np.testing.assert_array_equal(P[0], P[3])
    assert np.abs(P[0] - P[1]).max() > 0
    mgr.close()


@pytest.mark.parametrize("name,dtype,coupled", [("uniform_velocity", "f64", False), ("angular_rates", "f32", False), ("uniform_acceleration", "f64", True)])
def test_step_fused_on_a_batch_with_several_classes(models, name, dtype, coupled):
    """A second (Q, R) for a model joins the model's batch as a class; the temporally fused multi-tick call on such a batch is
    served tick by tick (there is no per-class fused kernel) with the results of single ticks -- it used to throw after the
    queue had already been flushed (round-2 advisor finding).  Against one oracle per class."""
    m = models[name]
    N, dt, ticks = 500, 0.004, 6
    rng = np.random.default_rng(21)
    Q, R, P0 = (_coupled_classes if coupled else _scaled_classes)(m, 2, rng)
    class_of = (np.arange(N) % 2).astype(np.uint32)
    ref = oracle.stream_fill(m["model"], 31, N, ticks, dt, dtype=dtype)
    from target_estimation_amd.streams import make_stream
    st = make_stream(m["model"], N, ticks, dt, 31, dtype=dtype)
    ids = np.arange(N, dtype=np.uint32)
    mgr = te.TargetManager(dtype=dtype)
    assert mgr.init_batch_classes(ids, dt, 0.0, ref["p0"], m["model"], Q, R, P0, class_of) == N
    b = mgr.batches()[0]
    assert len(mgr.batches()) == 1 and b.num_classes == 2
    mgr.update(3, dt, ref["meas"][0, 3])          # a queued one-target step must survive the call, too
    b.step_fused(dt, st["meas"])                  # 6 ticks "in one call": tick by tick for a multi-class batch
    members = {c: np.nonzero(class_of == c)[0] for c in (0, 1)}
    orcs = {}
    for c, rows in members.items():
        orc = oracle.OracleBatch(m["model"], Q[c], R[c], P0[c], ref["p0"][rows], dt, dtype=dtype)
        if c == 1:
            hit = np.zeros(len(rows), dtype=np.uint8); hit[list(rows).index(3)] = 1
            for i in np.nonzero(hit)[0]:
                import ctypes as C
                row = np.ascontiguousarray(ref["meas"][0, 3])
                orc._f("orc_target_add_measurement")(orc._at(int(i)), float(dt), row.ctypes.data_as(C.POINTER(C.c_double)))
        for s in range(ticks):
            orc.step(dt, ref["meas"][s][rows])
        orcs[c] = orc
    _check_classes(mgr, orcs, ids, members, dtype, "%s fused call on two classes" % name)
    assert mgr.getNumberMeasurements(0) == ticks and mgr.getNumberMeasurements(3) == ticks + 1
    mgr.close()
