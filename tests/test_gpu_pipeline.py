"""End-to-end: this package's run_dense_pipeline (HIP hot path) against the result upstream's
run_dense_pipeline produced on the same cameras / images / matcher outputs (tests/golden/g4)."""
import json
import os

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from lichtfeld_densification_plugin_amd.core.debug_viz import MatchDebugState
from helpers import oracle_cams

pytestmark = pytest.mark.gpu


class FakeMatcher:
    sample_thresh = 0.9

    def __init__(self, w, h, table, two_channel=False):
        self.w_resized, self.h_resized, self.table, self.calls, self.two = w, h, table, 0, two_channel
        self.closed = False

    def match_grids_batch(self, imA, imB_list):
        res = self.table[self.calls]
        self.calls += 1
        assert len(res) == len(imB_list)
        return [((w[..., 2:4].contiguous() if self.two else w), c) for w, c in res]

    def reference_axes(self, H, W):
        w0 = self.table[0][0][0]
        return w0[0, :, 0].contiguous(), w0[:, 0, 1].contiguous()

    def close(self):
        self.closed = True


def _scene(g4, tmp_path):
    from PIL import Image
    ocams = oracle_cams(g4)
    cams = []
    for i, c in enumerate(ocams):
        path = os.path.join(tmp_path, f"im{i:02d}.png")
        Image.fromarray(g4["images"][i]).save(path)
        cams.append(lfd.CameraRecord(uid=int(g4["cam_uid"][i]), image_path=path, width=c.width, height=c.height,
                                     K=c.K, R=c.R, t=c.t, P=c.P, C=c.C))
    refs = [int(r) for r in g4["refs_local"]]
    table = [[(torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j])) for j in range(2)] for r in refs]
    return cams, refs, g4["nn_table"], table


@pytest.mark.parametrize("mode,two_channel,backend", [("filter", False, "host"), ("nofilter", False, "host"),
                                                      ("filter", True, "host"), ("filter", False, "device"),
                                                      ("nofilter", False, "device")])
def test_pipeline_matches_upstream_run(g4, tmp_path, mode, two_channel, backend):
    """backend="host": the sampling stage makes upstream's own library calls -> identical run.
    backend="device": everything on the GPU; the only difference from upstream is the rounding of the
    weight normaliser (exact sum vs torch's thread-count-dependent f32 sum), which at this size leaves
    the drawn cells unchanged (tests/test_gpu_selection.py pins the draws themselves)."""
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = json.loads(str(g4[mode + "_cfg"]))
    cfg = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "out", "dense.ply"), roma_setting="fast",
                                  nns_per_ref=2, seed=5, viz_interval=2, pack_workers=1, selection_backend=backend, **kw)
    fm = FakeMatcher(64, 64, table, two_channel)
    progress, viz = [], []
    res = pl.run_dense_pipeline(cams, refs, nn, cfg, progress_callback=lambda p, m: progress.append((p, m)),
                                on_sequential_viz=lambda p: viz.append(os.path.basename(p)), matcher=fm)
    assert res.xyz.shape == g4[mode + "_xyz"].shape          # identical survivor count
    assert res.pairs_processed == int(g4[mode + "_pairs_processed"]) and res.pairs_matched == 6
    np.testing.assert_allclose(res.xyz, g4[mode + "_xyz"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res.rgb, g4[mode + "_rgb"], rtol=0, atol=1.0 / 255 / 4)
    if mode == "filter":
        np.testing.assert_allclose(res.err, g4[mode + "_err"], rtol=1e-5, atol=2e-3)
    np.testing.assert_array_equal([p for p, _ in progress], g4[mode + "_progress_pct"])
    assert [m.split(" | ")[0] for _, m in progress] == [str(m) for m in g4[mode + "_progress_msg"]]
    assert viz == [str(v) for v in g4[mode + "_viz_files"]]
    assert not fm.closed                                       # an injected matcher is the caller's to close
    assert res.xyz.dtype == np.float32 and res.rgb.dtype == np.float32 and res.err.dtype == np.float32


def test_pipeline_dense_mode_cancel_and_debug(g4, tmp_path):
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    cfg = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0,
                                  triangulation_mode="dense", refs_per_launch=2)
    res = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=FakeMatcher(64, 64, table))
    assert res.pairs_processed == 3 and res.xyz.shape[0] > 3 * 1000
    assert res.points_per_reference.sum() == res.xyz.shape[0]
    # cancellation between references
    calls = {"n": 0}

    def cancel():
        calls["n"] += 1
        return calls["n"] > 6

    with pytest.raises(pl.PipelineCancelled):
        pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=FakeMatcher(64, 64, table), cancel_requested=cancel)
    # debug previews carry survivors in match pixels
    st = MatchDebugState()
    st.set_enabled(True)
    cfg2 = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0,
                                   matches_per_ref=1200)
    pl.run_dense_pipeline(cams, refs, nn, cfg2, matcher=FakeMatcher(64, 64, table), debug_state=st)
    pv = st.latest()
    assert pv is not None and pv.matches.shape[1] == 4 and pv.matches.min() >= 0 and pv.matches.max() <= 63
    assert pv.cert_norm.min() >= 0 and pv.cert_norm.max() <= 1 and pv.total_pairs == 6
    # nothing survives -> upstream's RuntimeError text
    cfg3 = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, reproj_thresh=1e-9,
                                   matches_per_ref=1200)
    with pytest.raises(RuntimeError, match="No points triangulated. Try adjusting parameters."):
        pl.run_dense_pipeline(cams, refs, nn, cfg3, matcher=FakeMatcher(64, 64, table))


class _Node:
    def __init__(self, cam, focal_x, focal_y):
        self.has_camera, self.camera_uid = True, cam.uid
        self.camera_width, self.camera_height = cam.width, cam.height
        self.camera_focal_x, self.camera_focal_y = focal_x, focal_y
        self.camera_R, self.camera_T = cam.R, cam.t.reshape(3)
        self.image_path, self.has_mask, self.mask_path = cam.image_path, False, None


def test_dense_init_from_lfs_writes_the_same_ply_as_the_host_writer(g4, tmp_path):
    """GUI entry point end to end (camera nodes -> PLY): records packed on the device, cap applied,
    file byte-identical to the host writer's for the same result."""
    from lichtfeld_densification_plugin_amd import densify
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    nodes = [_Node(c, float(c.K[0, 0]), float(c.K[1, 1])) for c in cams]
    out = os.path.join(str(tmp_path), "gui", "dense.ply")
    cfg = lfd.DensePipelineConfig(output_path=out, nns_per_ref=2, num_refs=3, seed=5, viz_interval=0, matches_per_ref=1200,
                                  max_points=2000)
    msgs = []
    # nodes assume the principal point at the image centre, so geometry differs from g4: use a matcher that
    # replays the same maps for whatever references k-centres picks (3 of 6 cameras)
    class Replay(FakeMatcher):
        def match_grids_batch(self, imA, imB_list):
            res = self.table[self.calls % len(self.table)]
            self.calls += 1
            return [(res[j % len(res)][0], res[j % len(res)][1]) for j in range(len(imB_list))]
    code, info = densify.dense_init_from_lfs(nodes, cfg, progress_callback=lambda p, m: msgs.append((p, m)),
                                             matcher=Replay(64, 64, table))
    assert code == 0 and info == out and os.path.isfile(out)
    assert msgs[0] == (2.0, "Extracting camera data from scene...") and msgs[-1][0] == 100.0
    raw = open(out, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int(head.split(b"element vertex ")[1].split(b"\n")[0])
    assert 0 < n <= 2000 and len(body) == 15 * n
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    assert np.isfinite(rec["p"]).all()
    # the same run through the host writer gives the same bytes
    cfg2 = lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "gui", "host.ply"), nns_per_ref=2, num_refs=3,
                                   seed=5, viz_interval=0, matches_per_ref=1200, max_points=2000)
    recs = densify.extract_cameras_from_lfs(nodes)
    flat = np.stack([c.flat_pose() for c in recs])
    from lichtfeld_densification_plugin_amd.core.selection import nearest_neighbors, select_cameras_kcenters
    res = pl.run_dense_pipeline(recs, select_cameras_kcenters(flat, 3), nearest_neighbors(flat, 2), cfg2,
                                matcher=Replay(64, 64, table))
    x, c, e = densify._apply_point_cap(res.xyz, res.rgb, res.err, 2000, 5)
    writers.write_ply(cfg2.output_path, x, to_uint8_rgb(c))
    assert open(cfg2.output_path, "rb").read() == raw


def _rank_worker(rank, world, port, scene, cfg_kw, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks share the one GPU: host collectives
    try:
        cams, refs, nn, tables = scene
        mine = [tables[i] for i in range(len(refs)) if i % world == rank]
        cfg = lfd.DensePipelineConfig(**cfg_kw)
        res = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=FakeMatcher(64, 64, mine))
        q.put((rank, res.xyz, res.rgb, res.err, res.points_per_reference, res.pairs_processed, res.pairs_matched))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sampled", "dense"])
def test_two_ranks_reproduce_the_single_process_sequence(g4, tmp_path, mode):
    """References dealt round-robin to 2 ranks (each with its own HIP context) + ordered all-gather ==
    the 1-rank run with per-reference RNG streams: same points, same order."""
    import socket
    import torch.multiprocessing as mp
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200,
              triangulation_mode=mode, per_reference_rng=True)
    single = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, (cams, refs, nn, table), kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, xyz, rgb, err, counts, n_refs, n_pairs in results:
        np.testing.assert_array_equal(counts, single.points_per_reference)
        np.testing.assert_array_equal(xyz, single.xyz)
        np.testing.assert_array_equal(rgb, single.rgb)
        np.testing.assert_array_equal(err, single.err)
        assert n_refs == single.pairs_processed and n_pairs == single.pairs_matched


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two fresh rank processes itself (here both on the one GPU of the test box,
    LFD_BENCH_RANKS_PER_GPU=2, host collectives) and prints ONE line with n_gpus = 2.  The DEFAULT job of N > 1 is one scene (config 4's
    shape) dealt over the ranks - strong scaling - with the exchange of the survivors inside `value`; compute-only is a side key.  With fewer
    GPUs than ranks and no sharing it refuses instead of silently running one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFD_BENCH_RANKS_PER_GPU="2")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--refs", "6", "--preset", "turbo",
           "--light", "--spinup-s", "0.05"]
    res = subprocess.run(cmd + ["--replicate", "0"], env=env, capture_output=True, text=True, timeout=600)       # the pure-sharding schedule
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["replication"]["n_replicated"] == 0 and d["replication"]["forced"] and d["value"] == d["value_pure_sharding"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["refs_total"] == 6 and d["config"]["refs_per_gpu"] == 3 and d["config"]["neighbours"] == 8 and "config[3]" in d["config"]["workload"]
    assert d["rccl_ranks"] == 0 and d["collective_backend"] == "gloo"          # two ranks on one GPU exchange through host buffers, and the line says so
    ex = d["exchange"]
    assert ex["form"] == "all_gather" and ex["record_bytes"] == 15 and ex["overlapped"] and ex["rounds"] == 2 and ex["points"] > 0
    assert ex["bytes_gathered"] == 15 * ex["points"] and ex["end_of_run_28B"]["allgather_ms"] > 0 and ex["end_of_run_28B"]["gather_to_root_ms"] > 0
    assert "all_gather" in d["value_includes"] and 0 < d["value"] <= d["value_compute_only"] * 2.0         # the exchange is INSIDE value (three short steps of two processes on a shared host: the two figures are noisy, the bound is generous)
    assert d["compute_ms"] > 0 and d["kernel_ms"] > 0 and d["ms_per_step"] >= d["compute_ms"] * 0.5          # (two noisy figures of three short steps)
    assert d["value_sharded_resident"] > 0 and "counts_only" in d["value_sharded_resident_note"]         # the cloud left sharded: counts on the wire only
    assert d["sampled_mode"]["value"] > 0 and d["sampled_mode"]["points_per_scene"] > 0
    assert d["end_to_end"] is None and "unmeasured" in d["end_to_end_note"] and "no multi-GPU node" in d["scaling_note"]
    # recompute instead of communicate: the last two references computed by both ranks, four exchanged - the cloud is the pure-sharding one bit
    # for bit (bench.py asserts it) and every point counts once; then the plan from the run's own measurements (whatever it decides here)
    res_r = subprocess.run(cmd + ["--replicate", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert res_r.returncode == 0, res_r.stderr[-2000:]
    dr = json.loads([l for l in res_r.stdout.splitlines() if l.startswith("{")][0])
    rp = dr["replication"]
    assert rp["n_replicated"] == 2 and rp["n_sharded"] == 4 and rp["forced"] and rp["redundant_cells_per_step"] == 2 * dr["config"]["grid"][0] * dr["config"]["grid"][1]
    assert dr["exchange"]["points"] == ex["points"] and dr["exchange"]["rounds"] == 2 and dr["value"] > 0 and dr["value_pure_sharding"] > 0
    assert "2 references replicated" in dr["value_includes"] and "4 references round-robin" in dr["config"]["sharding"]
    res_a = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res_a.returncode == 0, res_a.stderr[-2000:]
    da = json.loads([l for l in res_a.stdout.splitlines() if l.startswith("{")][0])
    ra = da["replication"]
    assert not ra["forced"] and ra["n_replicated"] + ra["n_sharded"] == 6 and ra["inputs_measured_in_this_run"]["ref_ms"] > 0 and da["exchange"]["points"] == ex["points"]
    assert ra["inputs_measured_in_this_run"]["link_gbps"] > 0 and ra["planned_step_ms"] <= ra["planned_pure_sharding_ms"] + 1e-9
    cm = ra["candidates_measured_ms"]                     # "n replicated, r round(s)" -> ms per step
    reps = {int(k_.split()[0]) for k_ in cm}
    # pure sharding both ways is always measured; "everything replicated" never is (every rank would compute the whole scene and nothing would cross
    # a link: the headline would be the 1-GPU number, not strong scaling) - at least one reference per rank stays sharded
    assert 0 in reps and 6 not in reps and max(reps) <= 6 - 2 and ra["n_replicated"] in reps and "0 replicated, 1 round(s)" in cm and "0 replicated, 2 round(s)" in cm
    assert da["nothing_sharded"] is False
    assert min(v_ for k_, v_ in cm.items() if int(k_.split()[0]) == ra["n_replicated"]) == min(cm.values())      # measured, and the fastest one taken
    # the other forms: gather to the writer rank, 28-byte rows, weak scaling
    cmd_s = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--refs", "3", "--preset", "turbo",
             "--scaling", "weak", "--exchange", "gather_to_root", "--exchange-records", "f32", "--exchange-rounds", "3", "--light", "--spinup-s", "0.05",
             "--replicate", "0"]
    res_s = subprocess.run(cmd_s, env=env, capture_output=True, text=True, timeout=600)
    assert res_s.returncode == 0, res_s.stderr[-2000:]
    ds = json.loads([l for l in res_s.stdout.splitlines() if l.startswith("{")][0])
    res_g = subprocess.run(cmd_s[:-1] + ["3"], env=env, capture_output=True, text=True, timeout=600)       # ... with half of the scene replicated: on the root only
    assert res_g.returncode == 0, res_g.stderr[-2000:]
    dg = json.loads([l for l in res_g.stdout.splitlines() if l.startswith("{")][0])
    assert dg["replication"]["n_replicated"] == 3 and dg["replication"]["redundant_cells_per_step"] == 0 and dg["exchange"]["points"] == ds["exchange"]["points"]
    assert ds["scaling"] == "weak" and ds["config"]["refs_total"] == 6 and ds["config"]["refs_per_gpu"] == 3
    assert ds["exchange"]["form"] == "gather_to_root" and ds["exchange"]["record_bytes"] == 28 and ds["exchange"]["rounds"] == 3 and ds["value"] > 0
    if torch.cuda.device_count() < 2:
        env2 = dict(os.environ)
        env2.pop("WORLD_SIZE", None)
        env2.pop("LFD_BENCH_RANKS_PER_GPU", None)
        res2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=120)
        assert res2.returncode != 0 and "GPU(s) visible" in (res2.stderr + res2.stdout)


def test_bench_under_the_drivers_launcher(tmp_path):
    """the form the round-end driver uses: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` (here two ranks on
    the one GPU of the test box: LFD_BENCH_RANKS_PER_GPU=2 maps both local ranks to device 0 and the collectives to gloo): one JSON line from rank 0"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LFD_BENCH_RANKS_PER_GPU="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k_, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--refs", "6", "--preset", "turbo", "--light", "--spinup-s", "0.05"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong" and d["value"] > 0 and d["collective_backend"] == "gloo"
    assert d["replication"]["n_replicated"] + d["replication"]["n_sharded"] == 6 and d["host"]["torch_threads"] >= 1


@pytest.mark.parametrize("mode", ["sampled", "dense"])
def test_streamed_output_and_previews_are_the_host_writer_bytes(g4, tmp_path, mode):
    """config.stream_output: the PLY grows while the run proceeds (records packed on the device, appended per reference, vertex
    count patched into the header at the end) and equals upstream's write_ply of the final result byte for byte behind the
    header; every intermediate preview is a complete PLY of the points so far (upstream core/pipeline.py:508-532)."""
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    out_path = os.path.join(str(tmp_path), "s", "streamed.ply")
    cfg = lfd.DensePipelineConfig(output_path=out_path, nns_per_ref=2, seed=5, viz_interval=1, matches_per_ref=1200,
                                  triangulation_mode=mode, stream_output=True)
    previews = []
    res = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=FakeMatcher(64, 64, table),
                                on_sequential_viz=lambda p: previews.append(open(p, "rb").read()))
    assert res.streamed_path == out_path and res.xyz.shape[0] > 500
    raw = open(out_path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    ref_path = os.path.join(str(tmp_path), "ref.ply")
    writers.write_ply(ref_path, res.xyz, to_uint8_rgb(res.rgb))
    ref_head, ref_body = open(ref_path, "rb").read().split(b"end_header\n", 1)
    assert body == ref_body
    lines = [l for l in head.decode("ascii").split("\n") if l and not l.startswith("comment")]
    assert lines == [l for l in ref_head.decode("ascii").split("\n") if l]           # same header once the padding comment is dropped
    # previews: one per reference with points, each the complete cloud so far
    counts = np.cumsum([c for c in res.points_per_reference if c > 0])
    assert len(previews) == len(counts)
    for n, blob in zip(counts, previews):
        p = os.path.join(str(tmp_path), "prefix.ply")
        writers.write_ply(p, res.xyz[:n], to_uint8_rgb(res.rgb[:n]))
        assert blob == open(p, "rb").read()


@pytest.mark.parametrize("per_ref", [False, True])
def test_launch_ahead_equals_the_synchronous_loop(g4, tmp_path, per_ref, monkeypatch):
    """Sampled mode launches reference i+1 before it reads reference i back (the device's MT19937 stream advances in stream order,
    the counts come back behind an event): same points, same order, same per-reference counts as the one-at-a-time loop."""
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200,
              per_reference_rng=per_ref, upstream_normaliser=False,       # (upstream's normaliser needs the host between two references)
              refs_per_launch=1)                                          # (one reference per fused call: the automatic default would group them)
    ahead = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    calls = {"n": 0}
    orig = pl._HotPath.launch_sampled

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    monkeypatch.setattr(pl._HotPath, "launch_sampled", counted)
    again = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    assert calls["n"] == len(refs)                                   # the launch-ahead path is the one that ran
    monkeypatch.setattr(pl._HotPath, "can_launch_ahead", lambda self, *a, **k: False)
    sync = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    for r in (ahead, again):
        np.testing.assert_array_equal(r.xyz, sync.xyz)
        np.testing.assert_array_equal(r.rgb, sync.rgb)
        np.testing.assert_array_equal(r.err, sync.err)
        np.testing.assert_array_equal(r.points_per_reference, sync.points_per_reference)


@pytest.mark.parametrize("device_matcher", [False, True])
def test_device_image_prep_gives_the_host_prepared_run(g4, tmp_path, device_matcher):
    """config.device_image_prep: decoded images (97x130, not the match size) and "L" masks are resized / thresholded / blacked
    out on the GPU with Pillow's arithmetic instead of by PIL on the pack threads; the run is identical, whether the matcher
    takes the prepared images as device tensors or as PIL images."""
    from PIL import Image
    ocams = oracle_cams(g4)
    rs = np.random.RandomState(2)
    cams = []
    for i, c in enumerate(ocams):
        ip, mp = os.path.join(str(tmp_path), f"big{i:02d}.png"), os.path.join(str(tmp_path), f"big{i:02d}_mask.png")
        Image.fromarray(rs.randint(0, 256, (97, 130, 3)).astype(np.uint8)).save(ip)
        m = np.full((97, 130), 255, np.uint8)
        m[10 + i:40 + i, 20:70] = 0
        m[60:80, 90 + i:120] = 100                      # below the threshold
        Image.fromarray(m, mode="L").save(mp)
        cam = lfd.CameraRecord(uid=int(g4["cam_uid"][i]), image_path=ip, width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C)
        cam.mask_path = mp
        cams.append(cam)
    refs = [int(r) for r in g4["refs_local"]]
    table = [[(torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j])) for j in range(2)] for r in refs]
    seen = []

    class DevMatcher(FakeMatcher):
        accepts_device_images = True

        def match_grids_batch(self, imA, imB_list):
            seen.append((isinstance(imA, torch.Tensor) and imA.is_cuda and tuple(imA.shape) == (64, 64, 3),
                         all(isinstance(b, torch.Tensor) and b.is_cuda for b in imB_list)))
            return super().match_grids_batch(imA, imB_list)

    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200)
    host = pl.run_dense_pipeline(cams, refs, g4["nn_table"], lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    mk = DevMatcher if device_matcher else FakeMatcher
    devr = pl.run_dense_pipeline(cams, refs, g4["nn_table"], lfd.DensePipelineConfig(device_image_prep=True, **kw), matcher=mk(64, 64, table))
    assert host.xyz.shape[0] > 300
    np.testing.assert_array_equal(devr.xyz, host.xyz)
    np.testing.assert_array_equal(devr.rgb, host.rgb)
    np.testing.assert_array_equal(devr.err, host.err)
    np.testing.assert_array_equal(devr.points_per_reference, host.points_per_reference)
    if device_matcher:
        assert seen and all(a and b for a, b in seen)


def test_prepared_images_stay_on_the_device_once_per_camera(g4, tmp_path, monkeypatch):
    """device_image_prep prepares a camera ONCE per run: it appears as a reference and as a neighbour of others (here 3 cameras in
    3 x (1 + 2) places), the match-size tensors stay on the device (bounded by PREPARED_CACHE_BYTES, least recently used first
    out); with the cache switched off every appearance is prepared again - and the run is the same."""
    from PIL import Image
    ocams = oracle_cams(g4)
    rs = np.random.RandomState(4)
    cams = []
    for i, c in enumerate(ocams):
        ip, mp = os.path.join(str(tmp_path), f"p{i:02d}.png"), os.path.join(str(tmp_path), f"p{i:02d}_mask.png")
        Image.fromarray(rs.randint(0, 256, (90, 121, 3)).astype(np.uint8)).save(ip)
        m = np.full((90, 121), 255, np.uint8)
        m[5 + i:30 + i, 10:60] = 0
        Image.fromarray(m, mode="L").save(mp)
        cam = lfd.CameraRecord(uid=int(g4["cam_uid"][i]), image_path=ip, width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C)
        cam.mask_path = mp
        cams.append(cam)
    refs = [int(r) for r in g4["refs_local"]]
    table = [[(torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j])) for j in range(2)] for r in refs]
    calls = {"image": 0, "mask": 0}
    real_image, real_mask = hb.HipDensifier.prepare_image, hb.HipDensifier.prepare_mask

    def count_image(self, *a, **k):
        calls["image"] += 1
        return real_image(self, *a, **k)

    def count_mask(self, *a, **k):
        calls["mask"] += 1
        return real_mask(self, *a, **k)

    monkeypatch.setattr(hb.HipDensifier, "prepare_image", count_image)
    monkeypatch.setattr(hb.HipDensifier, "prepare_mask", count_mask)
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200, device_image_prep=True)
    cached = pl.run_dense_pipeline(cams, refs, g4["nn_table"], lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    n_places = len(refs) * 3
    used = {int(r) for r in refs} | {int(n) for r in refs for n in list(g4["nn_table"][r])[:2]}
    assert calls == {"image": len(used), "mask": len(used)} and len(used) < n_places
    calls.update(image=0, mask=0)
    from lichtfeld_densification_plugin_amd.core import hotpath
    monkeypatch.setattr(hotpath, "PREPARED_CACHE_BYTES", 0)
    plain = pl.run_dense_pipeline(cams, refs, g4["nn_table"], lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    assert calls == {"image": n_places, "mask": n_places}
    assert cached.xyz.shape[0] > 300
    np.testing.assert_array_equal(cached.xyz, plain.xyz)
    np.testing.assert_array_equal(cached.rgb, plain.rgb)
    np.testing.assert_array_equal(cached.points_per_reference, plain.points_per_reference)


def test_grouped_sampled_launches_equal_single_launches(g4, tmp_path):
    """per_reference_rng + refs_per_launch = 3: three references share one fused call (lfd_triangulate_sampled_multi: one aggregate
    launch, three selections each on its own MT19937 stream, one pair of indexed launches); same points, order and counts as one
    call per reference."""
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200,
              per_reference_rng=True)
    single = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=1, **kw), matcher=FakeMatcher(64, 64, table))
    for n in (2, 3):
        grouped = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=n, **kw), matcher=FakeMatcher(64, 64, table))
        np.testing.assert_array_equal(grouped.xyz, single.xyz)
        np.testing.assert_array_equal(grouped.rgb, single.rgb)
        np.testing.assert_array_equal(grouped.err, single.err)
        np.testing.assert_array_equal(grouped.points_per_reference, single.points_per_reference)
    assert single.xyz.shape[0] > 1000


# ---- N4 through the pipeline: the matcher mirror's feature sharing on an nn.Module model ---------------------------------------
class _IdDescriptor(torch.nn.Module):
    """Stand-in for romav2's DINOv3 Descriptor on the device: the "features" of an image are its 8x8 block means, from which the
    stub model below recovers WHICH camera the image belongs to - features served for the wrong camera change the pair matched."""

    def __init__(self):
        super().__init__()
        self.gain = torch.nn.Parameter(torch.ones(1), requires_grad=False)
        self.calls = 0

    def forward(self, img):
        self.calls += 1
        return [torch.nn.functional.adaptive_avg_pool2d(img, 8).reshape(img.shape[0], -1) * self.gain]


class _TableRoMa(torch.nn.Module):
    """RoMaV2's interface as core/matcher.py uses it (an nn.Module, ``f`` a registered child, ``self.f(img_B_lr)`` called from
    inside the model: RoMaV2/src/romav2/romav2.py:68,101,177); the match itself is looked up in upstream's captured maps by the
    camera pair the two feature vectors identify."""
    signatures = None      # (n_cams, 192) block means of the scene's images
    pair_table = None      # (cam_a, cam_b) -> (warp (H,W,4), cert (H,W))
    last = None

    class Cfg:
        def __init__(self, compile=False):
            pass

    def __init__(self, cfg=None):
        super().__init__()
        self.f = _IdDescriptor()
        self.H_lr = self.W_lr = 64
        self.H_hr = self.W_hr = None
        self.bidirectional = False
        _TableRoMa.last = self

    def apply_setting(self, s):
        pass

    def _load_image(self, im):
        dev = self.f.gain.device
        if isinstance(im, torch.Tensor):
            return (im.float() / 255.0 if im.dtype == torch.uint8 else im).to(dev)
        return (torch.from_numpy(np.array(im)).permute(2, 0, 1).unsqueeze(0).float() / 255.0).to(dev)

    def _camera_of(self, feats):
        sig = _TableRoMa.signatures.to(feats[0].device)
        return int(torch.cdist(feats[0], sig).argmin().item())

    def match_from_features(self, f_list_A, img_A_lr, imB, img_A_hr=None):
        img_b = torch.nn.functional.interpolate(self._load_image(imB), size=(self.H_lr, self.W_lr), mode="bicubic", align_corners=False,
                                                antialias=True)
        f_b = self.f(img_b)                                      # romav2.py:177
        warp, cert = _TableRoMa.pair_table[(self._camera_of(f_list_A), self._camera_of(f_b))]
        dev = img_b.device
        return {"warp_AB": warp[None, ..., 2:4].to(dev), "overlap_AB": cert[None, ..., None].to(dev)}


def _install_table_model(monkeypatch, g4, refs, nn):
    import sys
    import types
    imgs = torch.from_numpy(np.stack([g4["images"][i] for i in range(len(g4["images"]))])).permute(0, 3, 1, 2).float() / 255.0
    _TableRoMa.signatures = torch.nn.functional.adaptive_avg_pool2d(imgs, 8).reshape(imgs.shape[0], -1)
    _TableRoMa.pair_table = {(r, int(nn[r][j])): (torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j]))
                             for r in refs for j in range(2)}
    stub = types.ModuleType("romav2")
    stub.RoMaV2 = _TableRoMa
    monkeypatch.setitem(sys.modules, "romav2", stub)


@pytest.mark.parametrize("device_prep", [False, True])
def test_feature_sharing_on_an_nn_module_model_gives_the_same_run(g4, tmp_path, monkeypatch, device_prep):
    """The drop-in's default path (share_features=True) with the real matcher mirror on cuda:0 and an nn.Module model: every
    camera goes through the backbone once (PairSchedule.n_backbone_forwards_shared) instead of once per reference and once per
    pair, the model is left untouched, and the PipelineResult is the one of the unshared run and of the table-replaying fake."""
    from lichtfeld_densification_plugin_amd.core import matcher as mm
    from lichtfeld_densification_plugin_amd.core.scheduler import PairSchedule
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    _install_table_model(monkeypatch, g4, refs, nn)
    sched = PairSchedule(refs, nn, [c.uid for c in cams], 2)
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200,
              pack_workers=1, device_image_prep=device_prep)
    runs = {}
    for share in (True, False):
        m = mm.RomaMatcher(device="cuda:0", setting="fast")
        model = _TableRoMa.last
        child, state_keys = model.f, sorted(model.state_dict())
        runs[share] = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(share_features=share, **kw), matcher=m)
        assert model.f is child and sorted(model.state_dict()) == state_keys       # the vendored model is as it was
        assert getattr(m, "_feature_cache", None) is None                          # nothing of the run's features is kept
        assert child.calls == (sched.n_backbone_forwards_shared if share else sched.n_backbone_forwards_upstream)
        m.close()
    fake = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table, two_channel=True))
    assert sched.n_backbone_forwards_shared < sched.n_backbone_forwards_upstream and runs[True].xyz.shape[0] > 1000
    for other in (runs[False], fake):
        np.testing.assert_array_equal(runs[True].points_per_reference, other.points_per_reference)
        np.testing.assert_array_equal(runs[True].xyz, other.xyz)
        np.testing.assert_array_equal(runs[True].rgb, other.rgb)
        np.testing.assert_array_equal(runs[True].err, other.err)
    assert runs[True].pairs_matched == sched.n_pairs


def test_matcher_fakes_receive_camera_keys_and_the_cache_does_not_outlive_the_run(g4, tmp_path):
    """A matcher that declares ``supports_feature_keys`` is handed (reference index, [neighbour indices]) with every call while
    sharing is on, none while it is off - and a cache left over from an earlier run is always replaced (stale entries
    of another scene under the same camera keys)."""
    cams, refs, nn, table = _scene(g4, str(tmp_path))

    class KeyedFake(FakeMatcher):
        supports_feature_keys = True

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.cache_log, self.keys_seen = [], []

        def set_feature_cache(self, cache):
            self.cache_log.append(cache)

        def match_grids_batch(self, imA, imB_list, keys=None):
            self.keys_seen.append(keys)
            return super().match_grids_batch(imA, imB_list)

    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200)
    fm = KeyedFake(64, 64, table)
    fm.cache_log.append("stale cache of an earlier run")
    on = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=fm)
    assert fm.keys_seen == [(r, [int(n) for n in nn[r][:2]]) for r in refs]
    assert fm.cache_log[1] is not None and fm.cache_log[-1] is None and len(fm.cache_log[1]) == 0
    fm2 = KeyedFake(64, 64, table)
    fm2.cache_log.append("stale cache of an earlier run")
    off = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(share_features=False, **kw), matcher=fm2)
    assert fm2.keys_seen == [None] * len(refs) and fm2.cache_log[1:] == [None]
    np.testing.assert_array_equal(on.xyz, off.xyz)


def test_device_selection_with_upstreams_normaliser_is_the_host_stage_on_this_machine(tmp_path):
    """selection_backend="device" (the default) with upstream_normaliser (the default): the sampling weights are normalised with
    torch's own CPU f32 sum - upstream's library call, on this machine - and the draws run on the device.  Cells drawn, their order,
    the RNG stream across references and every output are those of selection_backend="host" (core/sampling.py = upstream's calls),
    at full size (512x512, M = 10000), where the exact sum and torch's differ on most maps."""
    from PIL import Image
    from lichtfeld_densification_plugin_amd import synthetic
    from lichtfeld_densification_plugin_amd.core.sampling import upstream_weight_sum
    H = W = 512
    cams = synthetic.ring_cameras(40, seed=0)
    refs = [2, 11, 23, 31]
    nn = np.array([synthetic.ring_neighbours(40, r, 3) for r in range(40)])
    table, differ = [], 0
    for gi, r in enumerate(refs):
        # tie-free certainties: among equal weights the order of upstream's coverage walk is NumPy's unstable argsort, which nothing reproduces
        s = synthetic.synth_reference(cams, r, list(nn[r]), H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=4, seed=70 + gi, cert_mode="tiefree")
        cams[r].image_path = os.path.join(str(tmp_path), f"v{r}.png")
        Image.fromarray(s.image.numpy()).save(cams[r].image_path)
        table.append([(s.warp[j], s.cert[j]) for j in range(3)])
        best = torch.clamp(s.cert, min=0.2).max(dim=0).values
        w = torch.clamp(best, max=0.9)[2:-2, 2:-2].double().sum().item()
        differ += int(np.float32(w) != np.float32(upstream_weight_sum(best)))
    for c in cams:
        if not c.image_path or not os.path.exists(c.image_path):
            c.image_path = cams[refs[0]].image_path                         # neighbours' images are only loaded, never sampled
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=3, seed=9, viz_interval=0, pack_workers=1)

    def run(**extra):
        return pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw, **extra), matcher=FakeMatcher(W, H, table))
    dev_up, host = run(selection_backend="device"), run(selection_backend="host")
    np.testing.assert_array_equal(dev_up.points_per_reference, host.points_per_reference)
    np.testing.assert_array_equal(dev_up.xyz, host.xyz)
    np.testing.assert_array_equal(dev_up.rgb, host.rgb)
    np.testing.assert_array_equal(dev_up.err, host.err)
    assert dev_up.xyz.shape[0] > 4 * 6000
    exact = run(selection_backend="device", upstream_normaliser=False)      # everything on the device: the exact sum
    assert abs(int(exact.xyz.shape[0]) - int(host.xyz.shape[0])) < 200
    print(f"[normaliser] torch's f32 sum differs from the exact sum on {differ} of {len(refs)} maps; "
          f"points: upstream's normaliser {dev_up.xyz.shape[0]}, exact sum {exact.xyz.shape[0]}")


@pytest.mark.parametrize("per_launch", [1, 2, 5])
def test_dense_streamer_writes_upstreams_file_from_kernel_made_records(g4, tmp_path, per_launch):
    """dense mode + stream_output with nobody else looking at the points (core/strategies.py::DensePlyStreamer): lfd_triangulate_dense_ply writes the
    15-byte records, they cross PCIe on a side stream into pinned double buffers, a writer thread appends them.  The file is upstream's write_ply of the
    plain dense run's result (core/writers.py:29-46 there), no f32 cloud is assembled, and the result's arrays - read back from the file when somebody
    asks - are that run's positions and quantised colours."""
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    from lichtfeld_densification_plugin_amd.core.stages import StageClock
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, triangulation_mode="dense", refs_per_launch=per_launch)
    plain = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "plain.ply"), **kw),
                                  matcher=FakeMatcher(64, 64, table))
    out_path = os.path.join(str(tmp_path), "s", "streamed.ply")
    clock = StageClock()
    res = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=out_path, stream_output=True, **kw),
                                matcher=FakeMatcher(64, 64, table), stage_clock=clock)
    assert res.streamed_path == out_path and res.device_points is None and res._arrays is None      # nothing but records so far
    assert res.n_points == plain.xyz.shape[0] and res.pairs_processed == plain.pairs_processed and res.pairs_matched == plain.pairs_matched
    np.testing.assert_array_equal(res.points_per_reference, plain.points_per_reference)
    ref_path = os.path.join(str(tmp_path), "ref.ply")
    writers.write_ply(ref_path, plain.xyz, to_uint8_rgb(plain.rgb))
    head, body = open(out_path, "rb").read().split(b"end_header\n", 1)
    ref_head, ref_body = open(ref_path, "rb").read().split(b"end_header\n", 1)
    assert body == ref_body
    assert [l for l in head.decode().split("\n") if l and not l.startswith("comment")] == [l for l in ref_head.decode().split("\n") if l]
    np.testing.assert_array_equal(res.xyz, plain.xyz)                          # (read back from the file)
    np.testing.assert_array_equal(to_uint8_rgb(res.rgb), to_uint8_rgb(plain.rgb))
    assert not res.err.any()
    st = res.stages
    assert st["d2h_bytes"] == 15 * res.n_points and st["kernel"]["calls"] >= 1 and st["write"]["seconds"] > 0 and st["match"]["calls"] == len(refs)


def test_stage_clock_attributes_a_sampled_run_without_changing_it(g4, tmp_path):
    """run_dense_pipeline(stage_clock=StageClock(sync=...)): the attribution run drains the device after every stage and takes the unfused calls so
    that `select` and `kernel` are separate stages - the points are the plain run's."""
    from lichtfeld_densification_plugin_amd.core.stages import StageClock
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(output_path=os.path.join(str(tmp_path), "o.ply"), nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200)
    plain = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table))
    assert plain.stages is None
    from lichtfeld_densification_plugin_amd.core import image_io
    image_io.load_rgb_u8.cache_clear()              # (the loaders' caches live in the process: the plain run has warmed them)
    clock = StageClock(sync=torch.cuda.synchronize)
    timed = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**kw), matcher=FakeMatcher(64, 64, table), stage_clock=clock)
    np.testing.assert_array_equal(timed.xyz, plain.xyz)
    np.testing.assert_array_equal(timed.rgb, plain.rgb)
    for stage in ("decode", "prepare", "match", "select", "kernel", "d2h"):
        assert timed.stages[stage]["seconds"] > 0 and timed.stages[stage]["calls"] >= 1, stage
    assert timed.stages["select"]["calls"] == len(refs) == timed.stages["kernel"]["calls"]


def test_dense_streamer_without_the_look_back_writes_every_references_point_set(g4, tmp_path):
    """experimental['dense_tile_segments'] + stream_output: the records come from lfd_triangulate_dense_ply_segments (no ordered retirement) and cross
    in one copy per reference: per reference the same 15-byte records as the ordered run's, in tile-retirement order; same counts, same header."""
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, triangulation_mode="dense", refs_per_launch=2, stream_output=True)
    a_path, b_path = os.path.join(str(tmp_path), "a.ply"), os.path.join(str(tmp_path), "b.ply")
    a = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=a_path, **kw), matcher=FakeMatcher(64, 64, table))
    b = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=b_path, experimental={"dense_tile_segments": True}, **kw),
                              matcher=FakeMatcher(64, 64, table))
    np.testing.assert_array_equal(a.points_per_reference, b.points_per_reference)
    ha, ba = open(a_path, "rb").read().split(b"end_header\n", 1)
    hb_, bb = open(b_path, "rb").read().split(b"end_header\n", 1)
    assert ha == hb_ and len(ba) == len(bb) == 15 * a.n_points
    ra, rb = np.frombuffer(ba, np.uint8).reshape(-1, 15), np.frombuffer(bb, np.uint8).reshape(-1, 15)
    offs = np.concatenate([[0], np.cumsum(a.points_per_reference)])
    for i in range(len(offs) - 1):
        sa, sb = ra[offs[i]:offs[i + 1]], rb[offs[i]:offs[i + 1]]
        np.testing.assert_array_equal(sa[np.lexsort(sa.T[::-1])], sb[np.lexsort(sb.T[::-1])])


def test_dense_streamer_failure_paths_raise_and_never_hang(g4, tmp_path, monkeypatch):
    """The writer thread of DensePlyStreamer: a failing append (disk full) is kept, the queued records are still drained, the run raises that error and the
    thread is gone; a cancellation between references raises PipelineCancelled with the thread stopped and the file closed (a valid PLY of what was written)."""
    import threading
    from lichtfeld_densification_plugin_amd.core import writers
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, triangulation_mode="dense", refs_per_launch=1, stream_output=True)
    real = writers.StreamedPlyWriter.append_packed
    calls = {"n": 0}

    def failing(self, body):
        calls["n"] += 1
        if calls["n"] == 2:
            raise OSError(28, "No space left on device")
        return real(self, body)
    monkeypatch.setattr(writers.StreamedPlyWriter, "append_packed", failing)
    with pytest.raises(OSError, match="No space left"):
        pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "full.ply"), **kw), matcher=FakeMatcher(64, 64, table))
    assert calls["n"] >= 2 and not [t for t in threading.enumerate() if t.name == "lfd-ply-writer"]
    monkeypatch.setattr(writers.StreamedPlyWriter, "append_packed", real)
    seen = {"n": 0}

    def cancel():
        seen["n"] += 1
        return seen["n"] > 5
    out = os.path.join(str(tmp_path), "cancelled.ply")
    with pytest.raises(pl.PipelineCancelled):
        pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=out, **kw), matcher=FakeMatcher(64, 64, table), cancel_requested=cancel)
    assert not [t for t in threading.enumerate() if t.name == "lfd-ply-writer"]
    head, body = open(out, "rb").read().split(b"end_header\n", 1)
    n = int([ln for ln in head.decode().split("\n") if ln.startswith("element vertex")][0].split()[-1])
    assert len(body) == 15 * n                                   # closed properly: the header counts what the body holds
    # ... and the next run on the same process is unaffected
    ok = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "ok.ply"), **kw), matcher=FakeMatcher(64, 64, table))
    assert ok.n_points > 3000


def test_dense_streamer_with_a_writer_that_falls_behind(g4, tmp_path, monkeypatch):
    """Six launches of one reference each, a file that is fast at first and then slower than the kernels: both buffer pairs are busy when a launch
    needs one - one of them still with a launch in flight.  (Round 5 found the first version waiting for exactly that pair without collecting its
    launch: a deadlock that only showed at `high` on a 2 GB file.)  The run completes and the file is the plain run's."""
    import time as _time
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    refs6, table6 = refs + refs, table + table
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, triangulation_mode="dense", refs_per_launch=1)
    plain = pl.run_dense_pipeline(cams, refs6, nn, lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "p.ply"), **kw), matcher=FakeMatcher(64, 64, table6))
    real = writers.StreamedPlyWriter.append_packed
    calls = {"n": 0}

    def slow(self, body):
        calls["n"] += 1
        if calls["n"] > 2:
            _time.sleep(0.25)
        return real(self, body)
    monkeypatch.setattr(writers.StreamedPlyWriter, "append_packed", slow)
    out = os.path.join(str(tmp_path), "slow.ply")
    res = pl.run_dense_pipeline(cams, refs6, nn, lfd.DensePipelineConfig(output_path=out, stream_output=True, **kw), matcher=FakeMatcher(64, 64, table6))
    assert calls["n"] == 6 and res.n_points == plain.xyz.shape[0]
    ref = os.path.join(str(tmp_path), "ref.ply")
    writers.write_ply(ref, plain.xyz, to_uint8_rgb(plain.rgb))
    assert open(out, "rb").read().split(b"end_header\n", 1)[1] == open(ref, "rb").read().split(b"end_header\n", 1)[1]


@pytest.mark.parametrize("seed", range(5))
def test_dense_streamer_under_random_writer_delays(g4, tmp_path, monkeypatch, seed):
    """nine launches' worth of references, 1-4 per launch, an appending file whose latency jumps around between 0 and 30 ms (seeded): whatever the
    interleaving of launches, copies and appends, the file is the plain run's"""
    import time as _time
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    rs = np.random.RandomState(seed)
    cams, refs, nn, table = _scene(g4, str(tmp_path))
    refs9, table9 = refs * 3, table * 3
    per_launch = int(rs.randint(1, 5))
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, triangulation_mode="dense", refs_per_launch=per_launch)
    plain = pl.run_dense_pipeline(cams, refs9, nn, lfd.DensePipelineConfig(output_path=os.path.join(str(tmp_path), "p.ply"), **kw), matcher=FakeMatcher(64, 64, table9))
    real = writers.StreamedPlyWriter.append_packed
    delays = rs.choice([0.0, 0.0, 0.002, 0.03], size=32)
    calls = {"n": 0}

    def jittery(self, body):
        _time.sleep(float(delays[calls["n"] % len(delays)]))
        calls["n"] += 1
        return real(self, body)
    monkeypatch.setattr(writers.StreamedPlyWriter, "append_packed", jittery)
    out = os.path.join(str(tmp_path), "j.ply")
    res = pl.run_dense_pipeline(cams, refs9, nn, lfd.DensePipelineConfig(output_path=out, stream_output=True, **kw), matcher=FakeMatcher(64, 64, table9))
    assert res.n_points == plain.xyz.shape[0] and calls["n"] == -(-9 // per_launch)
    ref = os.path.join(str(tmp_path), "ref.ply")
    writers.write_ply(ref, plain.xyz, to_uint8_rgb(plain.rgb))
    assert open(out, "rb").read().split(b"end_header\n", 1)[1] == open(ref, "rb").read().split(b"end_header\n", 1)[1]


# ---- several references per fused call on upstream's ONE stream (lfd_triangulate_sampled_chain) --------------------------------
def _chain_scene(d, n_cams=9, H=256, W=256, k=2):
    """nine cameras, 256 x 256 match grids (large enough for the multi-workgroup selection kernel, the only one that chains), one camera whose
    mask leaves fewer non-zero weights than draws: upstream's np.random.choice raises for that reference - before it draws"""
    from PIL import Image
    from lichtfeld_densification_plugin_amd import synthetic
    from lichtfeld_densification_plugin_amd.core import selection
    cams = synthetic.ring_cameras(n_cams, seed=77, arc=0.9)
    nn = selection.nearest_neighbors(np.stack([c.flat_pose() for c in cams]), k)
    for i, c in enumerate(cams):
        c.image_path = os.path.join(d, f"im{i:02d}.png")
        Image.fromarray(synthetic.synth_image(H, W, 40 + i).numpy()).save(c.image_path)
        c.mask_path = None
    blob = np.zeros((H, W), np.uint8)
    blob[H // 2, 10:60] = 255
    cams[4].mask_path = os.path.join(d, "mask04.png")
    Image.fromarray(blob, mode="L").save(cams[4].mask_path)
    refs = [0, 2, 3, 4, 5, 7, 8]
    table = []
    for r in refs:
        nbrs = [int(n) for n in nn[r][:k]]
        s = synthetic.synth_reference(cams, r, nbrs, H, W, W, H, noise_px=0.4, outlier_frac=0.05, channels=4, seed=500 + r, cert_mode="tiefree")
        table.append([(s.warp[j], s.cert[j]) for j in range(len(nbrs))])
    return cams, refs, nn, table, (W, H)


@pytest.mark.parametrize("upstream_normaliser", [True, False])
def test_chained_groups_on_one_stream_equal_the_plain_run(tmp_path, upstream_normaliser):
    """The default single-stream sampled mode with refs_per_launch = 2 / 4 / 16 (groups of 2+2+2+1, 4+3, 7): the same cloud, counts and order
    as one reference per call - which is upstream's sequence (fixtures g4 / g12) - including a reference whose selection is refused in the
    middle of a group (it logs upstream's error, draws nothing, and the references behind it are not affected)."""
    from fuzz_scenes import Table
    d = str(tmp_path)
    cams, refs, nn, table, size = _chain_scene(d)
    kw = dict(output_path=os.path.join(d, "o.ply"), nns_per_ref=2, seed=11, viz_interval=0, matches_per_ref=3000, use_masks=True,
              upstream_normaliser=upstream_normaliser)
    plain = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=1, **kw), matcher=Table(size[0], size[1], table))
    assert plain.xyz.shape[0] > 8000 and int(plain.points_per_reference[3]) == 0 and int((plain.points_per_reference > 0).sum()) == 6
    for n in (2, 4, 16):
        got = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=n, **kw), matcher=Table(size[0], size[1], table))
        np.testing.assert_array_equal(got.points_per_reference, plain.points_per_reference)
        np.testing.assert_array_equal(got.xyz, plain.xyz)
        np.testing.assert_array_equal(got.rgb, plain.rgb)
        np.testing.assert_array_equal(got.err, plain.err)
        assert (got.pairs_processed, got.pairs_matched) == (plain.pairs_processed, plain.pairs_matched)


def test_chained_groups_with_previews_and_a_debug_reference_in_between(tmp_path):
    """previews every second reference + a match-debug state that is switched on for the third reference only: that reference wants its
    aggregated map on the host and leaves the chained schedule (it runs alone, between two groups); the stream and the emission order stay
    upstream's"""
    from fuzz_scenes import Table
    d = str(tmp_path)
    cams, refs, nn, table, size = _chain_scene(d)
    kw = dict(nns_per_ref=2, seed=3, viz_interval=2, matches_per_ref=2000, use_masks=True)

    class Toggling(Table):
        def __init__(self, st, *a):
            super().__init__(*a)
            self.st = st

        def match_grids_batch(self, imA, imB_list, **k):
            res = super().match_grids_batch(imA, imB_list, **k)
            self.st.set_enabled(self.calls == 2)          # (the driver asks is_enabled() before it matches the NEXT reference)
            return res

    def run(n, sub):
        viz, seen = [], []
        st = MatchDebugState()
        submit = st.submit_preview
        st.submit_preview = lambda pv: (seen.append(int(pv.matches.shape[0])), submit(pv))
        os.makedirs(os.path.join(d, sub), exist_ok=True)
        cfg = lfd.DensePipelineConfig(refs_per_launch=n, output_path=os.path.join(d, sub, "o.ply"), **kw)
        res = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=Toggling(st, size[0], size[1], table), debug_state=st,
                                    on_sequential_viz=lambda p: viz.append((os.path.basename(p), open(p, "rb").read())))
        return res, viz, seen
    plain, pv, ps = run(1, "a")
    got, gv, gs = run(3, "b")
    np.testing.assert_array_equal(got.xyz, plain.xyz)
    np.testing.assert_array_equal(got.points_per_reference, plain.points_per_reference)
    assert [n for n, _ in gv] == [n for n, _ in pv] and len(pv) >= 2
    for (_, a), (_, b) in zip(gv, pv):
        assert a == b
    assert gs == ps


@pytest.mark.parametrize("status,group", [(5, 2), (4, 3), (5, 16)])
def test_a_void_grouped_call_is_redone_on_the_restored_stream(tmp_path, monkeypatch, status, group):
    """What the selection kernels report when a chain's bounded waits expire (5: nothing committed) or a reference is refused for inexactness (4: it
    drew nothing although upstream would have) cannot be provoked on a healthy device, so the status of ONE collected call is falsified here - the
    call itself, and the call launched behind it, have really run and really moved the device's stream.  The driver has to drop both, take the
    stream back to the checkpoint lfd_rng_checkpoint made before the first of them, and redo their references one at a time: the cloud is the
    plain run's, bit for bit - which it can only be if the stream was restored exactly."""
    from fuzz_scenes import Table
    from lichtfeld_densification_plugin_amd.core import hotpath
    d = str(tmp_path)
    cams, refs, nn, table, size = _chain_scene(d)
    kw = dict(output_path=os.path.join(d, "o.ply"), nns_per_ref=2, seed=11, viz_interval=0, matches_per_ref=3000, use_masks=True)
    plain = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=1, **kw), matcher=Table(size[0], size[1], table))
    real, seen, rolled = hotpath.HotPath.finish_sampled, [], []

    def falsified(self, handle, check_selection=True):
        res = real(self, handle, check_selection)
        if not check_selection:
            seen.append(len(res.sel_status))
            if len(seen) == 1:                          # the run's first grouped call
                res.sel_status[min(1, len(res.sel_status) - 1)] = status
        return res
    real_rollback = hotpath.HotPath.rollback_rng
    monkeypatch.setattr(hotpath.HotPath, "finish_sampled", falsified)
    monkeypatch.setattr(hotpath.HotPath, "rollback_rng", lambda self, place: (rolled.append(place), real_rollback(self, place))[1])
    got = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(refs_per_launch=group, **kw), matcher=Table(size[0], size[1], table))
    assert rolled == [0] and seen[0] == min(group, 7)
    np.testing.assert_array_equal(got.points_per_reference, plain.points_per_reference)
    np.testing.assert_array_equal(got.xyz, plain.xyz)
    np.testing.assert_array_equal(got.rgb, plain.rgb)
    np.testing.assert_array_equal(got.err, plain.err)
