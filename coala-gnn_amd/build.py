"""Build libcoala_hip.so (gfx950) in-tree with hipcc.  `python coala-gnn_amd/build.py [--force]`."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("coala_cache.hip", "coala_sampler.hip", "coala_host.cpp", "coala_coloring.cpp", "coala_comm.cpp")]
HEADERS = [os.path.join(_ROOT, "include", "coala_hip.h"), os.path.join(_HERE, "csrc", "coala_internal.h")]
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcoala_hip.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def sources():
    return [s for s in SOURCES if os.path.exists(s)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in sources() + HEADERS)


def build_lib(force=False, verbose=False):
    """Compile every HIP/C++ source of the product into one shared library for gfx950."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(_ROOT, "include")]
    cmd += sources() + ["-o", LIB_PATH + ".tmp", "-lrt", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
