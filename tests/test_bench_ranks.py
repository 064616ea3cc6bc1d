"""bench.py's rank start-up: `--gpus N` must really run N ranks (round-1 finding: the flag was parsed and ignored).

CPU: the dry run (no device work: spawn, rendezvous, the all-reduce that counts the ranks, the barrier-bracketed
repetitions, the JSON line).  GPU: the real bench, two ranks on the one GPU of the box over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_flag_starts_that_many_ranks():
    p, line = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["value"] is None and "dry-run" in line["data"]       # a dry run never reports a number
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1   # ONE line, from rank 0


def test_single_rank_dry_run():
    p, line = _run(["--steps", "8", "--warmup", "2", "--dry-run"])
    assert p.returncode == 0 and line["n_gpus"] == 1


def test_mismatch_between_flag_and_launcher_fails():
    # the launcher (here: a faked environment) started one rank, the flag says two
    p, line = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None
    assert "WORLD_SIZE=1" in p.stderr


def test_driver_style_launch_is_accepted():
    """The driver's own command for N > 1: torch.distributed.run starts the ranks, bench.py must not spawn again."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"]
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_world_size_8_dry_run_through_the_drivers_launcher():
    """The driver's N = 8 command shape with gloo and no device work: eight ranks rendezvous, the counting all-reduce sees eight,
    the barrier-bracketed repetitions run, ONE short line comes out of rank 0, and the gather rehearsal's collective completes."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29641", BENCH, "--gpus", "8", "--steps", "6", "--warmup", "2", "--dry-run"]
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["scaling"] == "weak" and line["value"] is None


def test_a_late_rank_costs_status_3_and_a_diagnostic_not_a_hang():
    """One of two ranks reaches the gather's collective entry later than the deadline allows: the punctual rank's watchdog ends
    it with status 3 and says why; the job ends (nobody waits for ever); the line was printed before and stands."""
    import time
    t0 = time.time()
    p, line = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--gather-deadline", "2"],
                   env={"TE_BENCH_TEST_LATE_RANK": "1:30"}, timeout=120)
    took = time.time() - t0
    assert p.returncode != 0
    assert "bench.py rank 0: pose gather still not finished" in p.stderr          # the punctual rank says why it leaves
    assert "bench.py rank 1: pose gather" not in p.stderr                         # the late one never got as far as the gather
    assert "exitcode: 3" in p.stderr                                              # ... and leaves with status 3 (the launcher's report)
    assert line is not None and line["n_gpus"] == 2          # the line went out first
    assert took < 60, took                                   # the late rank (30 s) was not waited for


def test_strong_scaling_rows_come_first_for_n_gt_1():
    sys.path.insert(0, ROOT)
    import bench
    names = bench.DEFAULT_EXTRA_MULTI.split(",")
    assert names[:3] == ["cfg4_1gpu_strong", "cfg4_1gpu32_strong", "cfg5_1gpu_strong"]
    first_weak = min(i for i, n in enumerate(names) if not n.endswith("_strong"))
    assert all(n.endswith("_strong") for n in names[:first_weak]) and first_weak >= 5
    for n in names:
        base = n[:-len("_strong")] if n.endswith("_strong") else n
        assert base in bench.WORKLOADS or base in bench.MIXED


def test_a_workload_much_slower_than_the_committed_record_is_reported(capsys):
    """bench.py compares every workload of a run with the newest committed side file and says so on stderr (and in the side
    file) when one takes more than 1.25 x its recorded time."""
    sys.path.insert(0, ROOT)
    import bench
    src, before = bench.previous_side_file()
    assert src is not None and src.startswith("profiles/") and "cfg4" in before and "cfg4_1gpu" in before
    slow = bench.slowdown_check({"cfg4": 2.0 * before["cfg4"], "cfg2": 1.1 * before["cfg2"], "not_in_the_record": 1.0})
    assert [e["name"] for e in slow] == ["cfg4"] and abs(slow[0]["ratio"] - 2.0) < 1e-9 and slow[0]["previous_record"] == src
    err = capsys.readouterr().err
    assert "WARNING: cfg4 takes" in err and "cfg2" not in err


def _fake_result():
    """A result record of the shape run_mixed() returns, with the longest strings bench.py can produce."""
    sys.path.insert(0, ROOT)
    import bench
    desc = max((v[0] for v in list(bench.WORKLOADS.values()) + list(bench.MIXED.values())), key=len)
    kern = "kf_step_sep_kernel<ModelAR,double,3>+query"
    res = dict(name="cfg5_1gpu64_replay", desc=desc, model="angular_rates+uniform_acceleration", dtype="f64", targets_per_gpu=1_000_000,
               layout="axis_separable_packed+axis_separable_packed", launch_mode="graph: one branch per batch, free-running inside blocks of ticks (replay only)",
               cycles_per_s=6.8421052631578947e9, ms_per_step=0.14615384615384616, achieved_gbs=6012.345678901234, state_bytes=408123456,
               residency="L3-assisted: state = 1.5 x the 256 MB Infinity Cache (zig-zag traversal reuses the part touched last)")
    dom = dict(kernel=kern, model="angular_rates", units_per_launch=500_000, algorithmic_bytes_per_unit=968, avg_launch_ms=0.0802345678,
               achieved_gbs=6031.23456789, traffic=484212345.678)
    return bench, res, dom


def test_the_line_is_short_enough_for_the_driver_to_parse():
    """Round 2's line was 30 KB and the driver's bounded stdout tail did not hold it (BENCH_r02.json: parsed = null).  The
    line now carries the contract's keys only; everything else goes to the side file."""
    import argparse
    bench, res, dom = _fake_result()
    args = argparse.Namespace(steps=20, warmup=5, side_file="")
    for world in (1, 8):
        out = bench.make_line(args, res, dom, world, {"_meta": {"commit": "0123456789ab"}})
        if world == 1:
            out["cpu_baseline"] = dict(value=2.9e6, unit="cycles/s", cores=16, kind="port",
                                       sample="10000 angular_rates x 12 ticks + 10000 angular_velocities x 31 ticks (f64, OpenMP static over "
                                              "targets, -O3 -march=native rebuilt on this host)")
            out["parity"] = bench.parity_summary(dict(targets_per_model=256, ticks=[1, 100, 1000], models={
                "angular_rates": dict(max_abs_x=[1e-15, 4.9e-14, 3e-14], max_rel_P=[1e-16, 4.3e-15, 2e-15], ids_exact=True),
                "angular_velocities": dict(max_abs_x=[1e-15, 1.1e-14, 1e-14], max_rel_P=[1e-16, 5.4e-15, 2e-15], ids_exact=True)}))
        line = json.dumps(bench.compact_line(out))
        assert len(line) + 1 < bench.LINE_LIMIT < 8192
        back = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in back
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_unit", "units_per_launch",
                  "avg_launch_ms", "traffic_source"):
            assert k in back["roofline"]
        assert "workload" in back["config"] and "side_file" in back["config"] and back["n_gpus"] == world
        assert "extra" not in back and "kernels" not in back["roofline"]
        if world == 1:
            assert set(back["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}


def test_an_oversized_line_sheds_optional_keys_instead_of_breaking():
    import argparse
    bench, res, dom = _fake_result()
    out = bench.make_line(argparse.Namespace(steps=20, warmup=5, side_file=""), res, dom, 1, {})
    out["parity"] = "x" * 6000
    line = bench.compact_line(out)
    assert "parity" not in line and len(json.dumps(line)) < bench.LINE_LIMIT and line["value"] == pytest.approx(res["cycles_per_s"], rel=1e-5)


def test_dry_run_lines_are_short():
    for a in (["--dry-run"], ["--gpus", "2", "--dry-run"]):
        p, line = _run(a + ["--steps", "4", "--warmup", "1"])
        assert p.returncode == 0
        assert all(len(ln) < 4096 for ln in p.stdout.splitlines())


@pytest.mark.gpu
def test_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """The literal `bench.py --gpus 2` (it starts its two ranks itself), gloo between them, both on the box's one GPU:
    every rank steps its own shard, rank 0 prints the one line (first) and writes the side file."""
    side = str(tmp_path / "side.json")
    p, line = _run(["--gpus", "2", "--steps", "8", "--warmup", "2", "--reps", "2", "--workload", "cfg4_64", "--extra-multi", "cfg4_1gpu_strong,uv1m_strong",
                    "--extra-steps", "4", "--side-file", side], env={"TE_BENCH_BACKEND": "gloo"}, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["targets_total"] == 2 * line["config"]["targets_per_gpu"]
    assert len(json.dumps(line)) < 4096 and "roofline" in line
    assert line["roofline"]["kernel"] == "kf_step_population_kernel<double>"       # two models per rank: one launch per tick
    extra = json.load(open(side))["extra"]
    # BASELINE configs[3] as stated -- 10^6 targets OVER the ranks, every model split evenly -- comes first
    assert extra[0]["name"] == "cfg4_1gpu_strong" and extra[0]["targets_per_gpu"] == 500_000 and extra[0]["n_gpus"] == 2
    assert extra[0]["cycles_per_s"] > extra[0]["targets_per_gpu"] * 2 / (extra[0]["ms_per_step"] * 1e-3) * 0.99    # whole-job rate
    assert extra[1]["name"] == "uv1m_strong" and extra[1]["targets_per_gpu"] == 500_000


@pytest.mark.gpu
def test_single_gpu_line_parses_and_carries_roofline_and_cpu_baseline(tmp_path):
    """The driver's command shape at N = 1 (fewer extras, to keep the test short): the last stdout line is the record."""
    side = str(tmp_path / "side.json")
    p, line = _run(["--steps", "6", "--warmup", "2", "--reps", "2", "--extra", "cfg2,ar4m64", "--extra-steps", "4", "--side-file", side], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 4096 and json.loads(last) == line
    assert line["config"]["name"] == "cfg4_1gpu" and line["dtype"] == "f64" and line["value"] > 1e7
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_unit"] * r["units_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    sidej = json.load(open(side))
    assert {e["name"] for e in sidej["extra"]} >= {"cfg2", "ar4m64"} and "configs0" in sidej and "parity" in sidej
    rows = sidej["configs0"]["rows"]
    assert [r["steps"] for r in rows] == [1000, 10000] and all(r["max_abs_pose_difference"] < 1e-9 for r in rows)


@pytest.mark.gpu
def test_extras_that_overrun_their_deadline_do_not_cost_the_line(tmp_path):
    """Everything after the line is optional: with a deadline the extras cannot meet, the line is still the last stdout line;
    the process says which extra it was in, marks the side file, and leaves with status 4 -- a stalled extra (a hang, a resident
    session that never ends) must not be recorded as a clean run."""
    side = str(tmp_path / "side.json")
    p, line = _run(["--workload", "cfg2", "--steps", "8", "--warmup", "2", "--reps", "2", "--no-cpu", "--extra", "ar4m64,av4m64,uv10m,ua10m", "--extra-steps", "64",
                    "--post-deadline", "0.05", "--side-file", side], timeout=600)
    assert p.returncode == 4, p.stderr[-3000:]
    assert line is not None and line["value"] > 0 and json.loads(p.stdout.strip().splitlines()[-1]) == line
    assert "did not finish within" in p.stderr and "running: ar4m64" in p.stderr
    sidej = json.load(open(side))
    assert sidej.get("extras_incomplete") is True and sidej.get("stalled_in") == "ar4m64"
