"""Every PAIR of settings of DensePipelineConfig either works together - and then gives the sequence of the plain run with the same semantics -
or is refused with a message (DensePipelineConfig.problem()): nothing is silently ignored.

The pairs are generated from the table below; the CPU tier runs them on the host backend (the CPU twin), ``-m gpu`` runs the pairs that involve
a device-only setting on the device.  What a sharded run adds (exchange form x mode x stream) is covered by tests/test_distributed_pipeline_cpu.py.
Reference: upstream core/config.py:7-26 (the 18 upstream fields), core/pipeline.py:783-928 (the loop every combination must reproduce)."""
import itertools
import os

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core import writers
from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
from helpers import oracle_cams

# option -> (field or experimental key, non-default value)
OPTIONS = {
    "dense": ("triangulation_mode", "dense"),
    "launch2": ("refs_per_launch", 2),
    "per_ref_rng": ("per_reference_rng", True),
    "select_host": ("selection_backend", "host"),
    "exact_sum": ("upstream_normaliser", False),
    "exact_colour": ("exact_colour", True),
    "stream": ("stream_output", True),
    "device_prep": ("device_image_prep", True),
    "no_share": ("share_features", False),
    "pairs2": ("pairs_per_forward", 2),
    "no_filter": ("no_filter", True),
    "cap": ("max_points", 100),
    "voxel": ("voxel_size", 0.05),
    "gather_root": ("exchange", "gather_to_root"),
    "x:segments": ("dense_tile_segments", True),
    "x:replicate": ("exchange_replicate", 0.5),
    "x:shared_file": ("stream_shared_file", True),
    "x:ply_records": ("exchange_records", "ply"),
    "x:no_overlap": ("exchange_overlap", False),
    "x:round": ("exchange_round", 2),
}
# settings that change WHICH points come out (everything else must leave the sequence alone)
SEMANTIC = ("dense", "no_filter", "per_ref_rng")
DEVICE_ONLY = ("device_prep", "x:segments", "launch2", "select_host", "exact_sum", "stream", "exact_colour")


def _kwargs(names, backend, out):
    kw = dict(output_path=out, nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200, pack_workers=1, backend=backend, experimental={})
    for n in names:
        field, value = OPTIONS[n]
        if n.startswith("x:"):
            kw["experimental"][field] = value
        else:
            kw[field] = value
    return kw


class _Replay:
    sample_thresh = 0.9

    def __init__(self, table):
        self.w_resized = self.h_resized = 64
        self.table, self.calls = table, 0

    def match_grids_batch(self, imA, imB_list):
        res = self.table[self.calls]
        self.calls += 1
        return res

    def close(self):
        pass


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    from PIL import Image
    from conftest import load_golden
    g4 = load_golden("g4_pipeline.npz")
    tmp = str(tmp_path_factory.mktemp("matrix"))
    cams = []
    for i, c in enumerate(oracle_cams(g4)):
        path = os.path.join(tmp, f"im{i:02d}.png")
        Image.fromarray(g4["images"][i]).save(path)
        cams.append(lfd.CameraRecord(uid=int(g4["cam_uid"][i]), image_path=path, width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C))
    refs = [int(r) for r in g4["refs_local"]]
    table = [[(torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j])) for j in range(2)] for r in refs]
    return dict(cams=cams, refs=refs, nn=g4["nn_table"], table=table, tmp=tmp, canon={})


def _run(scene, names, backend, tag):
    out = os.path.join(scene["tmp"], tag, "out.ply")
    cfg = lfd.DensePipelineConfig(**_kwargs(names, backend, out))
    return cfg, pl.run_dense_pipeline(scene["cams"], scene["refs"], scene["nn"], cfg, matcher=_Replay(scene["table"]))


def _check_pair(scene, a, b, backend):
    names = tuple(sorted({a, b}))
    kw = _kwargs(names, backend, os.path.join(scene["tmp"], "probe.ply"))
    probe = lfd.DensePipelineConfig(output_path="probe.ply")
    for k, v in kw.items():
        setattr(probe, k, v)
    why = probe.problem()
    if why is not None:
        # refused at construction, and by the driver when the fields were changed afterwards - with the same message
        if probe.problem() is not None:
            with pytest.raises(ValueError) as e1:
                lfd.DensePipelineConfig(**kw)
            assert str(e1.value) == why
        with pytest.raises(ValueError) as e2:
            pl.run_dense_pipeline(scene["cams"], scene["refs"], scene["nn"], probe, matcher=_Replay(scene["table"]))
        assert why in str(e2.value)
        return "refused"
    sem = tuple(n for n in names if n in SEMANTIC)
    key = (backend, sem) + (("exact_sum",) if backend == "device" and "exact_sum" in names else ())
    if key not in scene["canon"]:
        scene["canon"][key] = _run(scene, key[1] + key[2:], backend, "canon_" + "_".join(key[1] + key[2:]) + backend)[1]
    canon = scene["canon"][key]
    cfg, res = _run(scene, names, backend, "_".join(n.replace(":", "") for n in names) + backend)
    np.testing.assert_array_equal(res.points_per_reference, canon.points_per_reference)
    assert res.pairs_processed == canon.pairs_processed and res.pairs_matched == canon.pairs_matched
    np.testing.assert_array_equal(res.xyz, canon.xyz)
    if cfg.stream_output:
        assert res.streamed_path == cfg.output_path
        ref = os.path.join(scene["tmp"], "ref.ply")
        writers.write_ply(ref, canon.xyz, to_uint8_rgb(canon.rgb))
        assert open(cfg.output_path, "rb").read().split(b"end_header\n", 1)[1] == open(ref, "rb").read().split(b"end_header\n", 1)[1]
        np.testing.assert_array_equal(to_uint8_rgb(res.rgb), to_uint8_rgb(canon.rgb))
    else:
        assert res.streamed_path is None
        np.testing.assert_allclose(res.rgb, canon.rgb, rtol=0, atol=1e-6)          # exact_colour moves dense mode's f32 blend by <= 2.5e-7
        if "dense" not in names or ("exact_colour" in names) == ("exact_colour" in key[1]):
            np.testing.assert_array_equal(res.err, canon.err)
    return "ran"


PAIRS = list(itertools.combinations(sorted(OPTIONS), 2)) + [(n, n) for n in sorted(OPTIONS)]


def test_every_pair_of_settings_runs_as_the_plain_sequence_or_is_refused_with_a_message(scene):
    outcomes = {}
    for a, b in PAIRS:
        outcomes[(a, b)] = _check_pair(scene, a, b, "host")
    ran = sum(1 for v in outcomes.values() if v == "ran")
    refused = sum(1 for v in outcomes.values() if v == "refused")
    assert ran >= 100 and refused >= 40, (ran, refused)
    # a few verdicts spelled out, so that a rule that silently disappears is noticed
    assert outcomes[("cap", "stream")] == "refused" and outcomes[("stream", "voxel")] == "refused"
    assert outcomes[("device_prep", "device_prep")] == "refused"                   # on the host backend
    assert outcomes[("dense", "select_host")] == "refused" and outcomes[("dense", "x:segments")] == "refused"
    assert outcomes[("stream", "x:replicate")] == "refused" and outcomes[("x:shared_file", "x:shared_file")] == "refused"
    assert outcomes[("dense", "stream")] == "ran" and outcomes[("per_ref_rng", "stream")] == "ran"


def test_an_unknown_experimental_setting_is_an_error():
    with pytest.raises(ValueError, match="unknown experimental setting"):
        lfd.DensePipelineConfig(output_path="a.ply", experimental={"dense_tile_segment": True})
    cfg = lfd.DensePipelineConfig(output_path="a.ply")
    assert cfg.exp("exchange_overlap") is True and cfg.exp("exchange_round") == 0 and len(cfg.experimental) == 0


def test_upstreams_positional_construction_still_works():
    """upstream core/config.py:7-26: 18 fields in this order; the GUI panel constructs the dataclass with keywords, the CLI too"""
    import dataclasses
    names = [f.name for f in dataclasses.fields(lfd.DensePipelineConfig)]
    assert names[:18] == ["output_path", "roma_setting", "roi_only_selected", "num_refs", "nns_per_ref", "matches_per_ref", "certainty_thresh",
                          "reproj_thresh", "sampson_thresh", "min_parallax_deg", "max_points", "no_filter", "use_masks", "voxel_size", "seed",
                          "viz_interval", "prefetch_packages", "pack_workers"]
    assert len(names) == 31 and names[-1] == "experimental"
    cfg = lfd.DensePipelineConfig("o.ply", "fast", False, 0.8, 3, 10000, 0.2, 0.8, 5.0, 0.5, 0, False, True, 0.0, 0, 3, 8, 4)
    assert cfg.pack_workers == 4 and cfg.triangulation_mode == "sampled"


@pytest.mark.gpu
def test_pairs_with_a_device_only_setting_on_the_device(scene):
    """the GPU subset: every pair that involves a setting only the device backend implements (image preparation, the fused grouped call, the device
    selection and its normaliser, the unordered kernel, the records-only streamed output of dense mode)"""
    outcomes = {}
    for a, b in PAIRS:
        if a in DEVICE_ONLY or b in DEVICE_ONLY:
            outcomes[(a, b)] = _check_pair(scene, a, b, "device")
    ran = sum(1 for v in outcomes.values() if v == "ran")
    assert ran >= 70, (ran, len(outcomes))
    assert outcomes[("dense", "x:segments")] == "ran" and outcomes[("device_prep", "device_prep")] == "ran"
    # (round 5: several references per fused call ALSO on upstream's one stream - lfd_triangulate_sampled_chain - and it has to give the plain sequence)
    assert outcomes[("launch2", "per_ref_rng")] == "ran" and outcomes[("launch2", "launch2")] == "ran" and outcomes[("launch2", "select_host")] == "refused"
    assert outcomes[("dense", "stream")] == "ran"            # DensePlyStreamer


def test_automatic_references_per_launch():
    """refs_per_launch = 0 (the default) is resolved by the run: sixteen where the results do not depend on it and nobody waits for intermediate
    results, one otherwise; an explicit number is taken as it is."""
    from lichtfeld_densification_plugin_amd.core.types import AUTO_REFS_PER_LAUNCH
    cfg = lfd.DensePipelineConfig(output_path="a.ply")
    assert cfg.refs_per_launch == 0 and AUTO_REFS_PER_LAUNCH == 16
    assert cfg.launch_group() == 16 and cfg.launch_group(world=1, previews=False) == 16
    assert cfg.launch_group(previews=True) == 1                         # the GUI's previews keep upstream's cadence
    assert cfg.launch_group(world=2) == 1                               # every rank has to derive the same number from the configuration alone
    assert lfd.DensePipelineConfig(output_path="a.ply", backend="host").launch_group() == 1
    assert lfd.DensePipelineConfig(output_path="a.ply", selection_backend="host").launch_group() == 1
    assert lfd.DensePipelineConfig(output_path="a.ply", triangulation_mode="dense").launch_group() == 16
    assert lfd.DensePipelineConfig(output_path="a.ply", refs_per_launch=3).launch_group(previews=True) == 3
    with pytest.raises(ValueError, match="refs_per_launch"):
        lfd.DensePipelineConfig(output_path="a.ply", refs_per_launch=-1)
