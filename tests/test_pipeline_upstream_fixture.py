"""The driver loop on the CPU twin (``backend="host"``) against what UPSTREAM'S OWN run_dense_pipeline computed on the same seeded scenes
(tests/golden/g12_pipeline_upstream.npz, generated in the development container by tests/golden/make_pipeline_fixture.py from the imported reference:
core/pipeline.py:783-928 with its loader, `_collect_reference_matches`, `_triangulate_ref`).  The same fixture pins the DEVICE pipeline on the GPU box
(tests/test_gpu_pipeline_fuzz.py), where upstream's code cannot travel."""
import os

import pytest

from conftest import load_golden
from fuzz_scenes import N_SCENES, assert_is_upstreams_run, run, scene


@pytest.fixture(scope="module")
def g12():
    return load_golden("g12_pipeline_upstream.npz")


@pytest.mark.parametrize("sc", range(N_SCENES))
def test_cpu_twin_pipeline_is_upstreams_run(sc, tmp_path, g12):
    d = str(tmp_path)
    cams, refs, nn, table, size, kw = scene(sc, d)
    res, progress, viz = run(cams, refs, nn, table, size, os.path.join(d, "h", "dense.ply"), backend="host", **kw)
    assert_is_upstreams_run(res, progress, viz, g12, sc)


def test_the_fixture_covers_what_it_says(g12):
    assert int(g12["n_scenes"]) == N_SCENES
    pts = sum(int(g12[f"s{sc}_xyz"].shape[0]) for sc in range(N_SCENES) if f"s{sc}_xyz" in g12.files)
    errors = [sc for sc in range(N_SCENES) if f"s{sc}_error" in g12.files]
    assert pts > 30000 and len(errors) <= 3            # tens of thousands of upstream-made points; a scene or two where upstream raises
