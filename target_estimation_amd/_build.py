"""Build the HIP shared library in-tree (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEFAULT_LIB = os.path.join(HERE, "lib", "libtarget_estimation_amd.so")
TESTHOOKS_LIB = os.path.join(HERE, "lib", "libtarget_estimation_amd_testhooks.so")
# the library to load: the in-tree build, unless a deployment (or a test that needs the fault-injection build) names another
LIB = os.environ.get("TARGET_ESTIMATION_AMD_LIB") or DEFAULT_LIB


def build(force=False, jobs=8):
    """hipcc --offload-arch=gfx950 on every kernel / host source (csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(DEFAULT_LIB):
        raise RuntimeError("build did not produce %s" % DEFAULT_LIB)
    return DEFAULT_LIB


def build_testhooks(jobs=8):
    """The library with the fault-injection hooks (-DTE_TEST_HOOKS): test infrastructure, never loaded by default."""
    subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs, "testhooks"], stdout=subprocess.DEVNULL)
    return TESTHOOKS_LIB
