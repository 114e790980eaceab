"""Builds oracle/_ref/ from the reference's own sources WHERE THEY LIE (only in the build container, where
/root/reference exists).  Only the graph-colouring tool qualifies: graph_coloring.cpp compiles from its own two files with
g++ and the installed pybind11 / Python headers.  The cache path (CUDA + NVSHMEM + BaM) is unbuildable here.
Outputs stay out of git (.gitignore: oracle/_ref/) but travel to the GPU box with the snapshot."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/COALA_GNN_Modules"
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libref_coloring.so")


def build_ref(force=False):
    src = os.path.join(REF, "graph_coloring.cpp")
    if not os.path.exists(src):
        return None
    wrap = os.path.join(HERE, "ref_coloring_wrap.cpp")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > max(os.path.getmtime(wrap), os.path.getmtime(src)):
        return LIB
    import pybind11
    os.makedirs(OUT, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-I", REF, "-I", pybind11.get_include(),
           "-I", sysconfig.get_paths()["include"], wrap, src, "-o", LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_ref(force="--force" in sys.argv))
