"""Build the HIP shared library in-tree (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libtarget_estimation_amd.so")


def build(force=False, jobs=8):
    """hipcc --offload-arch=gfx950 on every kernel / host source (csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB):
        raise RuntimeError("build did not produce %s" % LIB)
    return LIB
