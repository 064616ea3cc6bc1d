"""CPU-only checks of host logic that needs no GPU (layout tables, model-file reader), compiled
with g++ from the product headers."""
import os
import subprocess

from conftest import MODEL_FILES, ROOT, model_path


def test_layout_tables_and_yaml_reader(tmp_path):
    exe = str(tmp_path / "host_logic_test")
    src = os.path.join(ROOT, "tests", "host", "host_logic_test.cpp")
    # AddressSanitizer + UBSan on the CPU build (the GPU pool has no sanitizer runs)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, src])
    out = subprocess.run([exe] + [model_path(k) for k in MODEL_FILES], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout


def test_device_quartic_solver_on_the_host_vs_oracle(tmp_path):
    """te_quartic.hpp (the solver the intersection kernels use) compiled for the host, against the oracle's
    long-double Aberth roots: sphere scenes down to |a| ~ 1e-12, random and prescribed-root quartics."""
    import oracle
    oracle.load()                                            # builds oracle/_build/libte_oracle.so on demand
    build = os.path.join(ROOT, "oracle", "_build")
    exe = str(tmp_path / "quartic_host_test")
    src = os.path.join(ROOT, "tests", "host", "quartic_host_test.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unknown-pragmas", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", exe, src, "-L", build, "-lte_oracle", "-Wl,-rpath," + build])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr
    assert "quartic host test ok" in out.stdout
