"""A plain C program using only the reference's ten C symbols, compiled with gcc against our header
and linked with the GPU library: the drop-in claim exercised from C, not through ctypes."""
import os
import subprocess

import pytest

from conftest import ROOT, model_path

pytestmark = pytest.mark.gpu


def test_plain_c_caller_of_the_ten_symbols(tmp_path):
    import torch   # the GPU box; without a device the program would (correctly) fail at target_manager_new
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "drop_in_test")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "tests", "c_abi", "drop_in_test.c"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-lm", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, model_path("uniform_velocity")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "drop-in test ok" in out.stdout
    assert "Target(7) already exists!" in out.stdout
    assert "Target(8) does not exist!" in out.stdout


def test_plain_c_caller_of_the_batch_extension(tmp_path):
    """tests/c_abi/batch_ext_test.c: gcc -std=c99 against target_batch_c.h -- parameter classes, by-id calls in random order
    (device-resolved), unknown ids, getters in the caller's order, erase, and the gather's argument validation."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "batch_ext_test")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "tests", "c_abi", "batch_ext_test.c"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-lm", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, model_path("uniform_acceleration")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "batch extension test ok" in out.stdout


def test_cpp_example_of_the_batch_api(tmp_path):
    """examples/batched_replay.cpp: the device-resident batch API driven from plain C++ (hipcc, no Python)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "batched_replay")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "examples", "batched_replay.cpp"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, model_path("angular_velocities"), "5000", "256"], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "5000 targets, 256 ticks" in out.stdout and "predict+update cycles/s" in out.stdout
    assert "512 measurements" in out.stdout          # 256 timed + 4 warm-up blocks of 64 ticks


@pytest.mark.parametrize("name,n,steps", [("uniform_velocity", 10000, 1000), ("angular_rates", 4000, 300)])
def test_cpp_example_of_the_resident_mode(tmp_path, name, n, steps):
    """examples/live_stream.cpp: the resident mode, the stream generator and the round-3 getters from plain C++ (hipcc, no
    Python): a session served through a ring that is refilled behind it equals the same ticks as single launches, bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "live_stream")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "examples", "live_stream.cpp"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, model_path(name), str(n), str(steps)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "live stream example ok" in out.stdout and "%d ticks served" % steps in out.stdout


@pytest.mark.parametrize("query", [False, True])
def test_cpp_example_of_a_mixed_population(tmp_path, query):
    """examples/mixed_population.cpp: BASELINE configs[3] / configs[4] from plain C++ -- two motion models in one manager, one
    launch per tick for both (target_manager_population_tick), the fused sphere query, a recorded graph, HIP-event timing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "mixed_population")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "examples", "mixed_population.cpp"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, os.path.join(ROOT, "models"), "6250", "128"] + (["query"] if query else []), capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mixed population example ok" in out.stdout and "ONE launch per tick" in out.stdout
    assert "6250 + 6250 targets" in out.stdout and "256 measurements per target" in out.stdout     # 2 warm-up + 2 timed blocks of 64
    if query:
        assert "intersections at the last tick" in out.stdout


@pytest.mark.parametrize("name,n", [("angular_velocities", 40), ("angular_rates", 700)])
def test_cpp_example_of_the_realtime_loop(tmp_path, name, n):
    """examples/realtime_loop.cpp: the reference's node loop without a copy or a launch per tick -- measurements stored through the
    PCIe BAR into the resident session's ring, poses written by the GPU into host-mapped memory, one doorbell per tick -- and the
    last poses equal a manager stepped by single launches on the same measurements, bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "realtime_loop")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "examples", "realtime_loop.cpp"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, model_path(name), str(n), "300"], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "realtime loop example ok" in out.stdout and "%d targets, 300 ticks" % n in out.stdout and "measurements-to-poses" in out.stdout


@pytest.mark.parametrize("name,n", [("angular_velocities", 40), ("uniform_acceleration", 300)])
def test_c_example_of_the_node_tick(tmp_path, name, n):
    """examples/node_tick.c: the reference node's tick as ONE by-id update call + ONE by-id getter call (the one-target queue and
    the getter table behind them at this size) equals the ten symbols called target by target on a second manager, bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    libdir = os.path.join(ROOT, "target_estimation_amd", "lib")
    exe = str(tmp_path / "node_tick")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "include", "target_estimation_amd"),
                           os.path.join(ROOT, "examples", "node_tick.c"), "-o", exe,
                           "-L", libdir, "-ltarget_estimation_amd", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, model_path(name), str(n), "200"], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "node tick ok" in out.stdout and "largest difference between the two managers' poses: 0;" in out.stdout
