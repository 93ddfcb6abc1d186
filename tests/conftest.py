import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): cross-compile it once if it is not there
    lib = os.path.join(ROOT, "lichtfeld-densification-plugin_amd", "liblfd_densify.so")
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "_lfd_build", os.path.join(ROOT, "lichtfeld-densification-plugin_amd", "csrc", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_geometry.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_selection.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_triangulate.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_writers.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_pipeline.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_selection_tables.npz")
