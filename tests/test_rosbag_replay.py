"""Recorded multi-target data (the reference's test/test_multiple_targets.bag, extracted to
tests/golden/multiple_targets_tf.npz) replayed through the ingest policy: on CPU with the policy
restatement + oracle, on the GPU with the product, compared tick by tick."""
import os

import numpy as np
import pytest

import oracle
from oracle.ingest_policy import RefIngest
from target_estimation_amd import rosbag_tf
from conftest import ROOT, model_path

FIX = os.path.join(ROOT, "tests", "golden", "multiple_targets_tf.npz")
BAG = "/root/reference/test/test_multiple_targets.bag"


def load_fixture():
    a = np.load(FIX)
    return [dict(recv_time=float(a["recv_time"][i]), stamp=float(a["stamp"][i]), frame_id=a["frame_id"][i].decode(),
                 child_frame_id=a["child_frame_id"][i].decode(), pose=a["pose"][i].copy()) for i in range(len(a["stamp"]))]


def test_reader_reproduces_the_fixture():
    if not os.path.exists(BAG):
        pytest.skip("reference tree not present (GPU box)")
    tr = rosbag_tf.read_tf(BAG)
    fx = load_fixture()
    assert len(tr) == len(fx) == 572
    for a, b in zip(tr, fx):
        assert a["child_frame_id"] == b["child_frame_id"] and a["frame_id"] == b["frame_id"]
        assert a["stamp"] == b["stamp"] and a["recv_time"] == b["recv_time"]
        np.testing.assert_array_equal(a["pose"], b["pose"])


def test_reader_agrees_with_the_bags_own_index():
    """Independent of the reader's walk over the chunks: the bag's INDEX section (written by the ROS recorder) declares how
    many messages the /tf connection holds; the reader must have decoded exactly that many, each TFMessage consumed to its
    last byte (parse_tf_message raises otherwise).  Stored in the fixture when it was made; re-derived here when the
    reference tree is present."""
    a = np.load(FIX)
    assert int(a["tf_messages_declared_by_the_bag_index"]) == int(a["tf_messages_decoded"]) == 572
    assert int(a["bag_conn_count"]) == 1 and int(a["bag_chunk_count"]) == 1
    if os.path.exists(BAG):
        declared, conns, chunks = rosbag_tf.declared_counts(BAG)
        stats = {}
        rosbag_tf.read_tf(BAG, stats=stats)
        assert declared == {"/tf": stats["messages"]} and (conns, chunks) == (1, 1)


def test_fixture_contents_and_cpu_replay(models):
    fx = load_fixture()
    names = sorted({t["child_frame_id"] for t in fx})
    assert names == ["target_0", "target_1", "target_2"]
    assert {t["frame_id"] for t in fx} == {"camera_depth_optical_frame"}
    q = np.array([t["pose"][3:] for t in fx])
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)
    m = models["angular_velocities"]
    ref = RefIngest(m["model"], m["Q"], m["R"], m["P"], expiration_time=2.0)
    seen = []
    rosbag_tf.replay(fx, ref, 1.0 / 50.0, on_tick=lambda k, t, res: seen.append(tuple(res[0].tolist())))
    assert (0, 1) in seen or (0, 1, 2) in seen
    assert any(2 in s for s in seen)
    # while target_0 is visible and recently measured, its filtered position stays close to the measurement
    last = {}
    for t in fx:
        last[t["child_frame_id"]] = t["pose"]
    ids, poses = ref.tick(1.0 / 50.0, fx[-1]["recv_time"])
    for i, p in zip(ids, poses):
        assert np.linalg.norm(p[:3] - last["target_%d" % i][:3]) < 0.25


@pytest.mark.gpu
def test_gpu_replay_matches_policy_oracle(models):
    te = pytest.importorskip("target_estimation_amd")
    fx = load_fixture()
    m = models["angular_velocities"]
    dt = 1.0 / 50.0
    mgr = te.TargetManager()
    ing = te.MeasurementIngest(mgr, m["model"], m["Q"], m["R"], m["P"], expiration_time=2.0)
    ref = RefIngest(m["model"], m["Q"], m["R"], m["P"], expiration_time=2.0)
    got, want = [], []
    n1 = rosbag_tf.replay(fx, ing, dt, on_tick=lambda k, t, res: got.append(res))
    n2 = rosbag_tf.replay(fx, ref, dt, on_tick=lambda k, t, res: want.append(res))
    assert n1 == n2 > 1000
    for (ids, poses), (rids, rposes) in zip(got, want):
        np.testing.assert_array_equal(ids, rids)
        np.testing.assert_allclose(poses, rposes, atol=1e-8)
    assert max(len(g[0]) for g in got) == 3
    ing.close(); mgr.close()
