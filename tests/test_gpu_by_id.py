"""The array-of-ids entry points with ids in ARBITRARY order at a size where they are resolved on the device
(csrc/id_resolve.hpp; >= 8192 ids): several batches in one manager, unknown ids, masks, an id named twice, getters,
and the table's rebuild after erase / create.  GPU vs the oracle (one oracle batch per model)."""
import numpy as np
import pytest

import oracle
from conftest import synth_stream
from test_gpu_parity import TOL

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def _cmp(mgr, ids, orc, dtype, what):
    t = TOL[dtype]
    x, P = mgr.get_state_batch(ids)
    xo, Po = orc.state()
    assert (np.abs(x - xo) - (t["x_atol"] + t["x_rtol"] * np.abs(xo))).max() <= 0, what
    assert (np.abs(P - Po) / np.abs(Po).max(axis=(1, 2), keepdims=True)).max() <= t["P_rel"], what


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_random_order_by_id_calls_resolved_on_device(models, dtype):
    names = ["uniform_velocity", "angular_rates"]
    N = [12000, 9000]
    dt, steps = 0.004, 5
    rng = np.random.default_rng(3)
    mgr = te.TargetManager(dtype=dtype)
    all_ids = rng.permutation(200000)[: sum(N)].astype(np.uint32)
    parts, base = [], 0
    for name, n in zip(names, N):
        m = models[name]
        p0, meas = synth_stream(name, n, steps, seed=5 + n)
        ids = all_ids[base:base + n]
        base += n
        assert mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"]) == n
        parts.append(dict(name=name, ids=ids, p0=p0, meas=meas, orc=oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype)))
    assert len(mgr.batches()) == 2
    ids_all = np.concatenate([p["ids"] for p in parts])
    unknown = np.arange(500000, 500050, dtype=np.uint32)
    for s in range(steps):
        meas_all = np.concatenate([p["meas"][s] for p in parts])
        order = rng.permutation(len(ids_all))
        take = order[: len(order) * 3 // 4] if s % 2 else order        # a subset on odd ticks
        call_ids = np.concatenate([ids_all[take], unknown])              # + ids that do not exist
        call_meas = np.concatenate([meas_all[take], np.zeros((len(unknown), 7))])
        has = (rng.random(len(call_ids)) < 0.8).astype(np.uint8) if s >= 2 else None
        got = mgr.update_batch(call_ids, dt, call_meas, has)
        assert got == len(take)
        # the same steps in the oracles
        off = 0
        for p in parts:
            n = len(p["ids"])
            stepped = np.zeros(n, dtype=bool); h = np.zeros(n, dtype=np.uint8)
            sel = take[(take >= off) & (take < off + n)] - off
            stepped[sel] = True
            if has is None:
                h[sel] = 1
            else:
                pos = {int(g): k for k, g in enumerate(take)}
                h[sel] = [has[pos[int(g + off)]] for g in sel]
            # targets not named this tick do not move: step only the named ones, one by one
            for i in np.nonzero(stepped)[0]:
                if h[i]:
                    p["orc"]._f("orc_target_add_measurement")(p["orc"]._at(int(i)), float(dt), oracle.oracle._dp(np.ascontiguousarray(p["meas"][s][i])))
                else:
                    p["orc"]._f("orc_target_update")(p["orc"]._at(int(i)), float(dt))
            off += n
    for p in parts:
        _cmp(mgr, p["ids"], p["orc"], dtype, p["name"])
    # getters by id, random order with unknown ids: rows of unknown ids are left as they were, found = False
    order = rng.permutation(len(ids_all))
    q = np.concatenate([ids_all[order][:15000], unknown])
    rng.shuffle(q)
    pose, twist, acc, found = mgr.get_est_batch(q)
    known = np.isin(q, ids_all)
    np.testing.assert_array_equal(found, known)
    assert np.isnan(pose[~known]).all() and np.isfinite(pose[known]).all()
    off = 0
    for p in parts:
        po = p["orc"].pose()
        lut = {int(i): k for k, i in enumerate(p["ids"])}
        rows = [r for r, v in enumerate(q) if int(v) in lut]
        want = np.array([po[lut[int(q[r])]] for r in rows])
        np.testing.assert_allclose(pose[rows], want, atol=TOL[dtype]["out_atol"])
    # an id named twice in one call = two consecutive steps (host path), same answer as two calls
    twice = np.concatenate([parts[0]["ids"][:9000], parts[0]["ids"][:10]])
    m2 = np.concatenate([parts[0]["meas"][0][:9000], parts[0]["meas"][1][:10]])
    assert mgr.update_batch(twice, dt, m2) == len(twice)
    o = parts[0]["orc"]
    for i in range(9000):
        o._f("orc_target_add_measurement")(o._at(i), float(dt), oracle.oracle._dp(np.ascontiguousarray(parts[0]["meas"][0][i])))
    for i in range(10):
        o._f("orc_target_add_measurement")(o._at(i), float(dt), oracle.oracle._dp(np.ascontiguousarray(parts[0]["meas"][1][i])))
    _cmp(mgr, parts[0]["ids"], o, dtype, "after a call that names ids twice")
    # erase + create: the device table is rebuilt; erased ids are unknown, moved records are found where they now live
    gone = parts[1]["ids"][::3]
    assert mgr.erase_batch(gone) == len(gone)
    new_ids = np.arange(600000, 600000 + 3000, dtype=np.uint32)
    m = models[names[0]]
    assert mgr.init_batch(new_ids, dt, 0.0, parts[0]["p0"][:3000], type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"]) == 3000
    q = np.concatenate([gone[:100], parts[1]["ids"][1::3], new_ids])
    rng.shuffle(q)
    assert mgr.update_batch(q, dt, None) == len(q) - 100          # predict-only by id
    pose, _, _, found = mgr.get_est_batch(q)
    np.testing.assert_array_equal(found, ~np.isin(q, gone))
    keep_rows = np.arange(len(parts[1]["ids"]))[1::3]
    for i in keep_rows:
        parts[1]["orc"]._f("orc_target_update")(parts[1]["orc"]._at(int(i)), float(dt))
    x, P = mgr.get_state_batch(parts[1]["ids"][1::3])
    xo, Po = parts[1]["orc"].state()
    t = TOL[dtype]
    assert (np.abs(x - xo[keep_rows]) - (t["x_atol"] + t["x_rtol"] * np.abs(xo[keep_rows]))).max() <= 0
    mgr.close()


def test_one_target_getters_sweeping_a_large_batch(models):
    """A reference-style caller at a scale the reference never reaches: 20 000 targets (more than the always-on getter
    table holds) read back ONE BY ONE through the ten-symbol getters.  The first reads after a change are single launches,
    a sweep switches to the host table (csrc/batch_store.hpp kBigDirect), one-target updates keep it current, a dense tick
    drops it -- at every stage each target's pose / twist / acceleration are the batched getter's, bit for bit."""
    name, dtype, N, dt = "uniform_acceleration", "f64", 20000, 0.004
    m = models[name]
    p0, meas = synth_stream(name, N, 4, seed=77)
    ids = (np.arange(N, dtype=np.uint32) * 3 + 1)
    mgr = te.TargetManager(dtype=dtype)
    assert mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"]) == N
    b = mgr.batches()[0]

    def sweep(which):
        pose, twist, acc, found = mgr.get_est_batch(ids)
        assert found.all()
        for k in which:
            i = int(ids[k])
            for got, want in ((mgr.getTargetPose(i), pose[k]), (mgr.getTargetTwist(i), twist[k]), (mgr.getTargetAcceleration(i), acc[k])):
                assert got[0]
                np.testing.assert_array_equal(got[1], want)

    mgr.update_batch(ids, dt, meas[0])
    sweep(range(0, N, 997))                       # a handful: single reads
    sweep(range(N))                               # a sweep: the table
    touched = np.arange(5, N, 401)
    for k in touched:                             # one-target updates behind a current table (the queue, then one indexed launch)
        mgr.update(int(ids[k]), dt, meas[1][k])
    sweep(list(touched) + list(range(0, N, 1999)))
    want_nm = np.ones(N, dtype=int); want_nm[touched] += 1
    assert [mgr.getNumberMeasurements(int(i)) for i in ids] == list(want_nm)       # the counters, target by target (host copy after a few reads)
    mgr.update(int(ids[3]), dt, meas[1][3]); want_nm[3] += 1                       # a step drops that copy
    assert [mgr.getNumberMeasurements(int(i)) for i in ids[:40]] == list(want_nm[:40])
    touched = np.append(touched, 3)
    b.step(dt, torch.from_numpy(np.ascontiguousarray(meas[2].T)).cuda())      # a dense tick: the table is stale
    sweep([0, N - 1])                             # (the batch swept last time: the table is rebuilt at once)
    sweep(range(N))
    b.step(dt, None)
    sweep([17])                                   # a single read after a change ...
    b.step(dt, None)
    sweep([N - 2, 3])                             # ... and the batch is back to single reads
    assert [mgr.getNumberMeasurements(int(i)) for i in ids[::13]] == list(want_nm[::13] + 1)     # the dense tick with measurements counted once more
    # and the oracle on the targets that took the extra one-target step
    sub = oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0[touched], dt, dtype=dtype)
    sub.step(dt, meas[0][touched]); sub.step(dt, meas[1][touched]); sub.step(dt, meas[2][touched]); sub.step(dt, None); sub.step(dt, None)
    _cmp(mgr, ids[touched], sub, dtype, "one-target updates of a large batch")
    mgr.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_node_tick_sized_calls_by_id(models, dtype):
    """Calls of up to 1024 ids take the one-target queue (TargetManager::updateBatch's small path): two models in one manager,
    random order, a subset, unknown ids, masks, an id named twice, predict-only, getters -- against the oracle, target by target."""
    names = ["angular_velocities", "uniform_acceleration"]
    N = [70, 45]
    dt, steps = 0.004, 7
    rng = np.random.default_rng(11)
    mgr = te.TargetManager(dtype=dtype)
    all_ids = rng.permutation(5000)[: sum(N)].astype(np.uint32)
    parts, base = [], 0
    for name, n in zip(names, N):
        m = models[name]
        p0, meas = synth_stream(name, n, steps + 1, seed=9 + n)
        ids = all_ids[base:base + n]
        base += n
        assert mgr.init_batch(ids, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"]) == n
        parts.append(dict(name=name, ids=ids, meas=meas, orc=oracle.OracleBatch(m["model"], m["Q"], m["R"], m["P"], p0, dt, dtype=dtype)))
    ids_all = np.concatenate([p["ids"] for p in parts])
    owner = np.concatenate([np.full(n, k) for k, n in enumerate(N)])
    local = np.concatenate([np.arange(n) for n in N])
    unknown = np.arange(9000, 9007, dtype=np.uint32)

    def oracle_step(g, s, with_meas):
        p = parts[owner[g]]
        i = int(local[g])
        if with_meas:
            p["orc"]._f("orc_target_add_measurement")(p["orc"]._at(i), float(dt), oracle.oracle._dp(np.ascontiguousarray(p["meas"][s][i])))
        else:
            p["orc"]._f("orc_target_update")(p["orc"]._at(i), float(dt))

    for s in range(steps):
        meas_all = np.concatenate([p["meas"][s] for p in parts])
        order = rng.permutation(len(ids_all))
        take = order[: len(order) * 2 // 3] if s % 2 else order
        if s == 3:
            take = np.concatenate([take, take[:5]])                      # five ids named twice: two consecutive steps each
        call_ids = np.concatenate([ids_all[take], unknown])
        call_meas = np.concatenate([meas_all[take], np.zeros((len(unknown), 7))])
        mix = rng.permutation(len(call_ids))
        call_ids, call_meas = call_ids[mix], call_meas[mix]
        has = (rng.random(len(call_ids)) < 0.7).astype(np.uint8) if s >= 2 else None
        predict_only = s == 5
        got = mgr.update_batch(call_ids, dt, None if predict_only else call_meas, None if predict_only else has)
        assert got == len(take)
        lut = {int(v): g for g, v in enumerate(ids_all)}
        for r, v in enumerate(call_ids):
            if int(v) in lut:
                oracle_step(lut[int(v)], s, (not predict_only) and (has is None or bool(has[r])))
        if s % 3 == 2:                                                   # getters between the ticks, unknown ids among them
            q = np.concatenate([ids_all[rng.permutation(len(ids_all))[:50]], unknown[:3]])
            rng.shuffle(q)
            pose, twist, acc, found = mgr.get_est_batch(q)
            known = np.isin(q, ids_all)
            np.testing.assert_array_equal(found, known)
            assert np.isnan(pose[~known]).all()
            for k, p in enumerate(parts):
                po, tw = p["orc"].pose(), p["orc"].twist()
                rows = [r for r, v in enumerate(q) if int(v) in lut and owner[lut[int(v)]] == k]
                sel = [int(local[lut[int(q[r])]]) for r in rows]
                np.testing.assert_allclose(pose[rows], po[sel], atol=TOL[dtype]["out_atol"])
                np.testing.assert_allclose(twist[rows], tw[sel], atol=TOL[dtype]["out_atol"] * 50, rtol=TOL[dtype]["x_rtol"] * 50)
    for p in parts:
        _cmp(mgr, p["ids"], p["orc"], dtype, p["name"])
    mgr.close()


def _bulk_path_case():
    """(child process, TE_SMALL_BATCH_QUEUE=0) the node-tick-sized test with the staged bulk paths of the same entry points"""
    import conftest
    models = {k: oracle.load_model_yaml(conftest.model_path(k)) for k in conftest.MODEL_FILES}
    for dtype in ("f64", "f32"):
        test_node_tick_sized_calls_by_id(models, dtype)
    print("bulk path ok")


def test_node_tick_sized_calls_through_the_bulk_paths():
    """The same calls with the queue path switched off (TE_SMALL_BATCH_QUEUE=0, read once per process: a child): the host look-up +
    indexed launch + staged copies that served these sizes until round 4 and still serve managers with a batch of more than
    16384 targets must give the oracle's answers too."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TE_SMALL_BATCH_QUEUE="0",
               PYTHONPATH=os.pathsep.join([os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__))]))
    p = subprocess.run([sys.executable, "-c", "import test_gpu_by_id as t; t._bulk_path_case()"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "bulk path ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
