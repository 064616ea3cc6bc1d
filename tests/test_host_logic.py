"""CPU-only checks of host logic that needs no GPU (layout tables, model-file reader), compiled
with g++ from the product headers."""
import os
import subprocess

from conftest import MODEL_FILES, ROOT, model_path


def test_layout_tables_and_yaml_reader(tmp_path):
    exe = str(tmp_path / "host_logic_test")
    src = os.path.join(ROOT, "tests", "host", "host_logic_test.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, src])
    out = subprocess.run([exe] + [model_path(k) for k in MODEL_FILES], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout
