"""The library's RCCL pose gather (csrc/pose_gather.cpp) on the one GPU of the box: a world-size-1 communicator
(ncclCommInitRank with one rank), two batches in the manager.  What can be checked here: the rows and their order,
and that a gather started BEFORE further ticks delivers the poses as they were at begin() -- the overlap contract.
(The multi-rank exchange itself needs more than one GPU; its protocol is covered by the gloo tests of dist.py.)"""
import numpy as np
import pytest

from conftest import synth_stream
from test_gpu_parity import to_soa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
te = pytest.importorskip("target_estimation_amd")


def test_gather_world_one_and_overlap(models):
    from target_estimation_amd.dist import PoseGather
    dt = 0.004
    mgr = te.TargetManager(dtype="f64")
    mgr.set_stream(torch.cuda.current_stream().cuda_stream)
    sizes = {"angular_rates": 5000, "uniform_acceleration": 3000}
    meas, base = {}, 0
    for name, n in sizes.items():
        m = models[name]
        p0, mm = synth_stream(name, n, 6, seed=2)
        mgr.init_batch(np.arange(n, dtype=np.uint32) + base, dt, 0.0, p0, type=m["model"], Q=m["Q"], R=m["R"], P0=m["P"])
        base += n
        meas[name] = mm
    batches = mgr.batches()
    names = list(sizes)
    for s in range(3):
        for b, name in zip(batches, names):
            b.step(dt, to_soa(meas[name][s], b))
    want = torch.cat([b.get_est(twist=False, acc=False)[0] for b in batches], 0).clone()
    g = PoseGather(mgr)
    assert g.world == 1 and g.counts() == [sum(sizes.values())]
    g.begin()
    for s in range(3, 6):                      # the next ticks are enqueued while the gather is in flight
        for b, name in zip(batches, names):
            b.step(dt, to_soa(meas[name][s], b))
    poses, ms = g.wait()
    torch.cuda.synchronize()
    assert poses.shape == (sum(sizes.values()), 7) and ms >= 0.0
    np.testing.assert_array_equal(poses.cpu().numpy(), want.cpu().numpy())     # the snapshot at begin(), not the later state
    later = torch.cat([b.get_est(twist=False, acc=False)[0] for b in batches], 0)
    assert (later - want).abs().max() > 0
    # a second round reuses the communicator and the buffers
    g.begin()
    poses2, _ = g.wait()
    np.testing.assert_array_equal(poses2.cpu().numpy(), later.cpu().numpy())
    # wrong counts are refused
    with pytest.raises(RuntimeError):
        g.begin(counts=[5])
    g.close()
    mgr.close()
