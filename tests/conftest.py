import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "coala-gnn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """libcoala_hip.so through the ctypes binding.  Built on demand with hipcc (cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("coala_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_lib()
    mod.build_pybind()
    import COALA_GNN_Pybind
    return COALA_GNN_Pybind
