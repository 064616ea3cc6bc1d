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
    """The library with the fault-injection hooks (-DTE_TEST_HOOKS): test infrastructure, never loaded by default.  A copy that
    is at least as new as the product library is taken as it is (both are built together in the authoring container and travel
    with the snapshot; the object files do not, and rebuilding every kernel on the GPU box costs minutes)."""
    if os.path.exists(TESTHOOKS_LIB) and os.path.exists(DEFAULT_LIB) and os.path.getmtime(TESTHOOKS_LIB) >= os.path.getmtime(DEFAULT_LIB):
        return TESTHOOKS_LIB
    subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs, "testhooks"], stdout=subprocess.DEVNULL)
    return TESTHOOKS_LIB
