"""Build libcoala_hip.so (gfx950) in-tree with hipcc.  `python coala-gnn_amd/build.py [--force]`."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("coala_cache.hip", "coala_sampler.hip", "coala_host.cpp", "coala_coloring.cpp", "coala_comm.cpp")]
PYBIND_SRC = os.path.join(_HERE, "csrc", "coala_pybind.cpp")
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


DEV_LIB_PATH = os.path.join(LIB_DIR, "libcoala_hip_dev.so")


def build_lib(force=False, verbose=False, dev=False):
    """Compile every HIP/C++ source of the product into one shared library for gfx950.
    dev=True: libcoala_hip_dev.so with -DCOALA_DEV_KNOBS (launch-geometry knobs from the environment, for tools/k1_insitu.py and
    friends; select it with COALA_HIP_LIB).  The product library reads no tuning knobs."""
    out = DEV_LIB_PATH if dev else LIB_PATH
    if dev and os.environ.get("COALA_EXTRA_HIPCC_FLAGS"):   # an experimental variant of the development build (-DK1_NT_STORES ...): its own file
        out = os.path.join(LIB_DIR, "libcoala_hip_var.so")
    if not force and not dev and not needs_build():
        return out
    # the development build is rebuilt only when a source is newer than it (or extra flags ask for another variant): the GPU tests ask for it
    # once per launch shape, and a compile of the whole library is ~50 s
    if dev and not force and not os.environ.get("COALA_EXTRA_HIPCC_FLAGS") and os.path.exists(out) and \
            os.path.getmtime(out) >= max(os.path.getmtime(p) for p in sources() + HEADERS + [os.path.abspath(__file__)]):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(_ROOT, "include"),
           # leading scalar kernel arguments arrive in SGPRs (gfx940+ kernarg preload): K1's first id load no longer waits for a
           # kernarg fetch (tools/k1_insitu.py: -0.15 us per launch)
           "-mllvm", "-amdgpu-kernarg-preload-count=16"]
    if dev:
        cmd.append("-DCOALA_DEV_KNOBS")
        cmd += os.environ.get("COALA_EXTRA_HIPCC_FLAGS", "").split()
    cmd += sources() + ["-o", out + ".tmp", "-lrt", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def pybind_path():
    import sysconfig
    return os.path.join(_HERE, "COALA_GNN_Pybind", "_coala_pybind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pybind(force=False, verbose=False):
    """Compile the pybind11 module COALA_GNN_Pybind/_coala_pybind (host C++ only, g++) against libcoala_hip.so -- the compiled
    counterpart of the reference's COALA_GNN_Pybind.cu.  Needs the library built first."""
    import pybind11
    import sysconfig
    out = pybind_path()
    deps = [PYBIND_SRC] + HEADERS
    if not force and os.path.exists(out) and all(os.path.getmtime(p) <= os.path.getmtime(out) for p in deps):
        return out
    build_lib()
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I", pybind11.get_include(),
           "-I", sysconfig.get_paths()["include"], PYBIND_SRC, "-o", out + ".tmp",
           "-L", LIB_DIR, "-lcoala_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
    if "--dev" not in sys.argv:
        print(build_pybind(force="--force" in sys.argv, verbose=True))
