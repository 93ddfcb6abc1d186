"""The whole driver loop sharded over two ranks WITHOUT a GPU: run_dense_pipeline(backend="host") (the CPU twin) under gloo,
references dealt round-robin, both exchange modes and the streamed writer of a sharded run (BASELINE config 5's "streamed writer"),
against the single-process run and upstream's write_ply of its result."""
import os
import socket

import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import pipeline as pl
from helpers import oracle_cams


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


def _scene(g4, tmp):
    from PIL import Image
    cams = []
    for i, c in enumerate(oracle_cams(g4)):
        path = os.path.join(tmp, f"im{i:02d}.png")
        if not os.path.exists(path):
            Image.fromarray(g4["images"][i]).save(path)
        cams.append(lfd.CameraRecord(uid=int(g4["cam_uid"][i]), image_path=path, width=c.width, height=c.height, K=c.K, R=c.R, t=c.t, P=c.P, C=c.C))
    refs = [int(r) for r in g4["refs_local"]]
    table = [[(torch.from_numpy(g4[f"ref{r}_warp"][j]), torch.from_numpy(g4[f"ref{r}_cert"][j])) for j in range(2)] for r in refs]
    return cams, refs, g4["nn_table"], table


def _rank(rank, world, port, tmp, cfg_kw, q):
    import torch.distributed as dist
    from conftest import load_golden
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, refs, nn, table = _scene(load_golden("g4_pipeline.npz"), tmp)
        mine = [table[i] for i in range(len(refs)) if i % world == rank]
        res = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**cfg_kw), matcher=_Replay(mine))
        q.put((rank, res.xyz, res.rgb, res.err, res.points_per_reference, res.pairs_processed, res.pairs_matched, res.streamed_path))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,exchange,form", [("sampled", "all_gather", "stream"), ("sampled", "gather_to_root", "stream"),
                                                ("dense", "gather_to_root", "stream"), ("dense", "all_gather", "shared_file"),
                                                ("dense", "gather_to_root", "shared_file"), ("sampled", "all_gather", "rounds"),
                                                ("dense", "gather_to_root", "rounds"), ("sampled", "gather_to_root", "end"), ("dense", "all_gather", "end")])
def test_two_host_ranks_equal_one_and_stream_the_same_file(g4, tmp_path, mode, exchange, form):
    """``form``: what the sharded run exchanges (core/sinks.py::ShardLink) - "stream": the streamed file, records sent to rank 0, then one exchange of
    the result; "shared_file": the streamed file written by both ranks into their own byte ranges (only counts travel); "rounds": the exchange in
    rounds beside the compute (OverlappedExchange, rounds of ONE reference here so that several rounds run); "end": the one exchange after the last
    reference"""
    import torch.multiprocessing as mp
    from lichtfeld_densification_plugin_amd.core import writers
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    tmp = str(tmp_path)
    cams, refs, nn, table = _scene(g4, tmp)
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200, triangulation_mode=mode, per_reference_rng=True, backend="host",
              pack_workers=1)
    single = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=os.path.join(tmp, "single.ply"), **kw), matcher=_Replay(table))
    writers.write_ply(os.path.join(tmp, "single.ply"), single.xyz, to_uint8_rgb(single.rgb))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = os.path.join(tmp, "sharded.ply")
    streams = form in ("stream", "shared_file")
    cfg_kw = dict(output_path=out, exchange=exchange, stream_output=streams,
                  experimental={"exchange_overlap": form != "end", "exchange_round": 1, "stream_shared_file": form == "shared_file"}, **kw)
    procs = [ctx.Process(target=_rank, args=(r, 2, port, tmp, cfg_kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    world = 2
    for rank, xyz, rgb, err, counts, n_refs, n_pairs, streamed in results:
        np.testing.assert_array_equal(counts, single.points_per_reference)
        assert n_refs == single.pairs_processed and n_pairs == single.pairs_matched and streamed == (out if streams else None)
        if exchange == "all_gather" or rank == 0:
            np.testing.assert_array_equal(xyz, single.xyz)
            np.testing.assert_array_equal(rgb, single.rgb)
            np.testing.assert_array_equal(err, single.err)
        else:                                   # gather_to_root: the other rank returns its own shard
            offs = np.concatenate([[0], np.cumsum(single.points_per_reference)])
            own = np.concatenate([single.xyz[offs[g]:offs[g + 1]] for g in range(rank, len(refs), world)])
            np.testing.assert_array_equal(xyz, own)
    if not streams:
        return
    # the streamed file of the sharded run: the single run's cloud through upstream's writer, byte for byte behind the header
    head, body = open(out, "rb").read().split(b"end_header\n", 1)
    ref_head, ref_body = open(os.path.join(tmp, "single.ply"), "rb").read().split(b"end_header\n", 1)
    assert body == ref_body
    assert [l for l in head.decode().split("\n") if l and not l.startswith("comment")] == [l for l in ref_head.decode().split("\n") if l]


def _failing_rank(rank, world, port, tmp, cfg_kw, q):
    import torch.distributed as dist
    from conftest import load_golden
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, refs, nn, table = _scene(load_golden("g4_pipeline.npz"), tmp)
        mine = [table[i] for i in range(len(refs)) if i % world == rank]
        pl.has_cached_romav2_weights = lambda: True

        def make_matcher(**_kw):          # the pipeline builds its own matcher (matcher=None): rank 1's constructor fails
            if rank == 1:
                raise RuntimeError("no model on this rank")
            return _Replay(mine)
        pl.RomaMatcher = make_matcher
        try:
            pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**cfg_kw))
            q.put((rank, None))
        except Exception as exc:
            q.put((rank, f"{type(exc).__name__}: {exc}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_a_rank_that_fails_before_the_loop_does_not_hang_the_others(g4, tmp_path, overlap):
    """with a streamed sharded output, a rank whose matcher cannot even be constructed used to reach `finally` without a stream
    object and went straight to the status agreement while rank 0 waited for its records.  The stream (``overlap`` False) or the overlapped
    exchange (True) now exists before anything can fail: the failing rank sends empty references / closes empty rounds, everybody raises."""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    _scene(g4, tmp)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cfg_kw = dict(output_path=os.path.join(tmp, "sharded.ply"), stream_output=not overlap, experimental={"exchange_round": 1}, nns_per_ref=2, seed=5,
                  viz_interval=0, matches_per_ref=1200, triangulation_mode="sampled", per_reference_rng=True, backend="host", pack_workers=1)
    procs = [ctx.Process(target=_failing_rank, args=(r, 2, port, tmp, cfg_kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] == "RuntimeError: no model on this rank"
    assert results[0] == "RuntimeError: dense pipeline failed on another rank"


def _rank_replicating(rank, world, port, tmp, cfg_kw, q):
    import torch.distributed as dist
    from conftest import load_golden
    from lichtfeld_densification_plugin_amd.core import distributed as lfd_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, refs, nn, table = _scene(load_golden("g4_pipeline.npz"), tmp)
        cfg = lfd.DensePipelineConfig(**cfg_kw)
        n_rep = int(round(cfg.exp("exchange_replicate") * len(refs)))
        positions, _n_sh = lfd_dist.split_replicated(len(refs), n_rep, rank, world, replicas_here=(cfg.exchange == "all_gather" or rank == 0))
        res = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=_Replay([table[g] for g in positions]))       # the matcher sees exactly these references
        q.put((rank, res.xyz, res.rgb, res.err, res.points_per_reference, res.pairs_processed, res.pairs_matched, len(positions)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,exchange,records,frac", [("dense", "all_gather", "f32", 0.5), ("sampled", "gather_to_root", "f32", 0.34), ("dense", "all_gather", "ply", 1.0),
                                                        ("sampled", "all_gather", "f32", 0.5)])
def test_replicated_references_give_the_single_process_result(g4, tmp_path, mode, exchange, records, frac):
    """experimental['exchange_replicate']: the last references of the list are computed by every rank that receives the cloud and never sent; the others are
    sharded and exchanged in rounds.  Same sequence, same counts, every reference and pair counted once."""
    import torch.multiprocessing as mp
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    tmp = str(tmp_path)
    cams, refs, nn, table = _scene(g4, tmp)
    kw = dict(nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200, triangulation_mode=mode, per_reference_rng=True, backend="host", pack_workers=1)
    single = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(output_path=os.path.join(tmp, "single.ply"), **kw), matcher=_Replay(table))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cfg_kw = dict(output_path=os.path.join(tmp, "sharded.ply"), exchange=exchange,
                  experimental={"exchange_round": 1, "exchange_records": records, "exchange_replicate": frac}, **kw)
    procs = [ctx.Process(target=_rank_replicating, args=(r, 2, port, tmp, cfg_kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_rep = int(round(frac * len(refs)))
    assert n_rep >= 1
    for rank, xyz, rgb, err, counts, n_refs, n_pairs, n_mine in results:
        assert n_refs == single.pairs_processed and n_pairs == single.pairs_matched            # replicated references and their pairs count once
        if exchange == "all_gather" or rank == 0:
            assert n_mine == len(range(rank, len(refs) - n_rep, 2)) + n_rep
            np.testing.assert_array_equal(counts, single.points_per_reference)
            np.testing.assert_array_equal(xyz, single.xyz)
            if records == "ply":                                                               # 15-byte records: colours as the writer quantises them, no error column
                np.testing.assert_array_equal(to_uint8_rgb(rgb), to_uint8_rgb(single.rgb))
            else:
                np.testing.assert_array_equal(rgb, single.rgb)
                np.testing.assert_array_equal(err, single.err)
        else:
            assert n_mine == len(range(rank, len(refs) - n_rep, 2))                            # not a consumer of the cloud: its share of the sharded part only
            np.testing.assert_array_equal(counts[:len(refs) - n_rep], single.points_per_reference[:len(refs) - n_rep])


def _rank_repeating(rank, world, port, tmp, cfg_kw, q):
    import torch.distributed as dist
    from torch.distributed import distributed_c10d as c10d
    from conftest import load_golden
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, refs, nn, table = _scene(load_golden("g4_pipeline.npz"), tmp)
        mine = [table[i] for i in range(len(refs)) if i % world == rank]
        groups, counts = [], []
        for _ in range(3):
            res = pl.run_dense_pipeline(cams, refs, nn, lfd.DensePipelineConfig(**cfg_kw), matcher=_Replay(mine))
            groups.append(len(c10d._world.pg_map))
            counts.append(int(res.points_per_reference.sum()))
        q.put((rank, groups, counts))
    finally:
        dist.destroy_process_group()


def test_repeated_streamed_sharded_runs_leave_no_process_group_behind(g4, tmp_path):
    """a streamed sharded run creates a process group of its own (dist.new_group()); it is destroyed at the end of the run - three runs in one
    process (the plugin lives in LichtFeld Studio for hours) leave exactly the default group, and each gives the same cloud"""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    _scene(g4, tmp)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cfg_kw = dict(output_path=os.path.join(tmp, "sharded.ply"), stream_output=True, nns_per_ref=2, seed=5, viz_interval=0, matches_per_ref=1200,
                  triangulation_mode="dense", per_reference_rng=True, backend="host", pack_workers=1)
    procs = [ctx.Process(target=_rank_repeating, args=(r, 2, port, tmp, cfg_kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, groups, counts in results:
        assert groups == [groups[0]] * 3 and groups[0] == 1, (rank, groups)         # only the default group is left after every run
        assert counts == [counts[0]] * 3 and counts[0] > 3000
