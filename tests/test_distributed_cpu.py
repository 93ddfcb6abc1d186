"""world_size-2 gloo tests of the sharding + ordered all-gather (CPU, runs without a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lichtfeld_densification_plugin_amd.core import distributed as lfd_dist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_points(n_refs, seed=0):
    rs = np.random.RandomState(seed)
    counts = rs.randint(0, 40, size=n_refs)
    counts[2 % n_refs] = 0
    pts = [rs.normal(size=(c, 7)).astype(np.float32) + 100.0 * g for g, c in enumerate(counts)]
    return counts, pts


def _worker(rank, world, port, n_refs, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        local = np.concatenate([pts[g] for g in mine], 0) if mine else np.zeros((0, 7), np.float32)
        t = torch.from_numpy(local)
        gx, gc, ge, gcounts = lfd_dist.all_gather_by_reference(t[:, 0:3], t[:, 3:6], t[:, 6], [counts[g] for g in mine],
                                                                n_refs, dist)
        ax, ac, ae, rc = lfd_dist.all_gather_points(t[:, 0:3], t[:, 3:6], t[:, 6], dist)
        # the same exchange as a gather to rank 0: the root holds the ordered cloud, the other rank keeps its own shard
        rx, rcol, re, rcounts = lfd_dist.gather_to_root_by_reference(t[:, 0:3], t[:, 3:6], t[:, 6], [counts[g] for g in mine], n_refs, dist)
        root = np.concatenate([rx.numpy(), rcol.numpy(), re.numpy()[:, None]], 1)
        # streamed PLY: 15-byte records per reference (here: the first 15 bytes of every 28-byte record), rank 0 writes the file
        path = os.path.join(tmp, "stream.ply")
        from lichtfeld_densification_plugin_amd.core.writers import StreamedPlyWriter
        writer = StreamedPlyWriter(path) if rank == 0 else None
        stream = lfd_dist.ShardedPlyStream(dist, n_refs, writer, torch.device("cpu"))
        for i, g in enumerate(mine):
            if counts[g]:                         # references without points are never pushed (the pipeline's emit() is not called)
                stream.push(i, torch.from_numpy(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        stream.finish()
        if writer is not None:
            writer.close()
        dist.barrier()
        q.put((rank, gx.numpy(), gc.numpy(), ge.numpy(), gcounts, ax.numpy(), rc, root, rcounts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs,world", [(7, 2), (2, 2), (1, 2), (10, 3), (5, 4), (2, 3)])
def test_ordered_all_gather_world2(n_refs, world, tmp_path):
    """(world 3 and 4: uneven shards, a rank without any reference when n_refs < world ... )"""
    tmp = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_refs, q, tmp)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs)
    full = np.concatenate(pts, 0)
    for rank, gx, gc, ge, gcounts, ax, rc, root, rcounts in results:
        np.testing.assert_array_equal(rcounts, counts)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        own = np.concatenate([pts[g] for g in mine], 0) if mine else np.zeros((0, 7), np.float32)
        np.testing.assert_array_equal(root, full if rank == 0 else own)     # gather_to_root: the ordered cloud on rank 0 only
        np.testing.assert_array_equal(gcounts, counts)
        np.testing.assert_array_equal(gx, full[:, 0:3])      # == the single-process sequence
        np.testing.assert_array_equal(gc, full[:, 3:6])
        np.testing.assert_array_equal(ge, full[:, 6])
        assert sum(rc) == full.shape[0] and ax.shape[0] == full.shape[0]


def test_shard_references_partition():
    for n, w in ((56, 8), (7, 2), (3, 4), (0, 2)):
        owned = [lfd_dist.shard_references(n, r, w) for r in range(w)]
        assert sorted(sum(owned, [])) == list(range(n))
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def test_streamed_file_of_two_ranks_is_the_single_writer_file(tmp_path):
    """ShardedPlyStream (exercised by the workers above): the file rank 0 wrote while both ranks pushed their references equals
    what one process appending the references in order writes."""
    from lichtfeld_densification_plugin_amd.core.writers import StreamedPlyWriter
    n_refs, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_refs, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for _ in range(world):
        q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs)
    ref = os.path.join(str(tmp_path), "single.ply")
    with StreamedPlyWriter(ref) as w:
        for g in range(n_refs):
            if counts[g]:
                w.append_packed(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).tobytes())
    assert open(os.path.join(str(tmp_path), "stream.ply"), "rb").read() == open(ref, "rb").read()


# ---- the exchange in rounds beside the compute (OverlappedExchange) ----------------------------------------------------------------------
def _overlap_worker(rank, world, port, n_refs, per_round, form, record, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs, seed=3)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        rec_dist = lfd_dist.RecordingDist(dist)        # forwards every call, logs the collectives (the dry run's plan is compared with this log)
        ex = lfd_dist.OverlappedExchange(rec_dist, n_refs, per_round, torch.device("cpu"), form=form, record=record)
        for i, g in enumerate(mine):
            if not counts[g]:
                continue                                  # a reference without survivors is never pushed
            t = torch.from_numpy(pts[g])
            ex.push(i, t if record == "f32" else torch.from_numpy(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        recs, gcounts = ex.finish()
        q.put((rank, recs.numpy(), gcounts, ex.n_rounds, rec_dist.log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs,world,per_round,form,record", [
    (7, 2, 1, "all_gather", "f32"), (7, 2, 2, "gather_to_root", "f32"), (10, 3, 2, "all_gather", "ply"), (5, 4, 1, "gather_to_root", "ply"),
    (2, 3, 4, "all_gather", "f32"), (1, 2, 1, "gather_to_root", "f32"), (9, 2, 16, "all_gather", "ply")])
def test_overlapped_exchange_is_the_single_process_sequence(n_refs, world, per_round, form, record):
    """world 2 / 3 / 4, rounds of 1, 2, 4 and "everything in one round", uneven shards, a reference without survivors, a rank without any
    reference: all_gather leaves the 1-rank sequence on every rank, gather_to_root on rank 0 (the others keep their shard)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, n_refs, per_round, form, record, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs, seed=3)

    def as_sent(parts):
        if not parts:
            return np.zeros((0, 7), np.float32) if record == "f32" else np.zeros((0,), np.uint8)
        full = np.concatenate(parts, 0)
        return full if record == "f32" else np.ascontiguousarray(full.view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)
    # the dry run (core/distributed.py::exchange_schedule: no communicator, no GPU) names the collectives this run really issued, on every rank, in
    # order: op, elements handed over and received, dtype - what RCCL will match operation by operation on the first real multi-GPU run
    plan = lfd_dist.exchange_schedule(n_refs, world, per_round, form, record, counts=counts)
    for rank, recs, gcounts, n_rounds, log in results:
        np.testing.assert_array_equal(gcounts, counts)
        assert n_rounds == -(-(-(-n_refs // world)) // per_round) == plan["rounds"]
        mine = lfd_dist.shard_references(n_refs, rank, world)
        assert plan["ranks"][rank]["positions"] == mine
        expect = as_sent(pts) if (form == "all_gather" or rank == 0) else as_sent([pts[g] for g in mine])
        np.testing.assert_array_equal(recs, expect)
        issued = [e for e in log if e["op"] in ("all_gather_into_tensor", "gather")]
        assert len(issued) == len(plan["collectives"]), (rank, issued, plan["collectives"])
        for got, want in zip(issued, plan["collectives"]):
            assert got["op"] == want["op"] and got["numel_in"] == want["numel_in"] and got["dtype"] == want["dtype"], (rank, got, want)
            if want["op"] == "all_gather_into_tensor":
                assert got["numel_out"] == want["numel_out"]
            else:
                assert got["dst"] == want["dst"]


def test_ply_records_give_back_positions_and_quantised_colours():
    rs = np.random.RandomState(0)
    xyz = rs.normal(size=(50, 3)).astype(np.float32)
    rgb = rs.uniform(size=(50, 3)).astype(np.float32)
    from lichtfeld_densification_plugin_amd.core.image_io import to_uint8_rgb
    from lichtfeld_densification_plugin_amd.core.writers import ply_records
    rec = torch.from_numpy(ply_records(xyz, to_uint8_rgb(rgb)).view(np.uint8).reshape(-1).copy())
    x, c = lfd_dist.points_from_ply_records(rec)
    np.testing.assert_array_equal(x.numpy(), xyz)
    np.testing.assert_array_equal(to_uint8_rgb(c.numpy()), to_uint8_rgb(rgb))          # re-quantising gives the same bytes


def test_backend_name_does_not_invent_a_backend():
    """a group that was never initialised is an error, not "nccl" (which would send host tensors down the device branch)"""
    assert not dist.is_initialized()
    with pytest.raises(Exception):
        lfd_dist._backend_name(dist)


# ---- failure paths of the streamed sharded writer -----------------------------------------------------------------------------------------
class _FailingWriter:
    def __init__(self, fail_at):
        self.calls, self.fail_at, self.kept = 0, fail_at, []

    def append_packed(self, data):
        self.calls += 1
        if self.calls == self.fail_at:
            raise OSError("disk full")
        self.kept.append(bytes(data))


def _writer_failure_worker(rank, world, port, n_refs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        writer = _FailingWriter(fail_at=2) if rank == 0 else None
        stream = lfd_dist.ShardedPlyStream(dist, n_refs, writer, torch.device("cpu"))
        raised = None
        for i, g in enumerate(mine):
            if counts[g]:
                stream.push(i, torch.from_numpy(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        try:
            stream.finish()
        except OSError as exc:
            raised = str(exc)
        dist.barrier()                       # both ranks get here: nobody is left waiting for a receive that never comes
        q.put((rank, raised, writer.calls if writer else None))
    finally:
        dist.destroy_process_group()


def test_a_writer_that_fails_on_rank_0_drains_the_peers_and_raises_afterwards():
    n_refs, world = 9, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_writer_failure_worker, args=(r, world, port, n_refs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == "disk full" and results[0][2] == 2          # the error, once; the writer was not called again
    assert results[1][1] is None and results[2][1] is None


# ---- exchange-free streamed output: only counts travel, every rank writes its own byte ranges of the file ---------------------------------
def _shared_file_worker(rank, world, port, n_refs, per_round, path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs, seed=3)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        st = lfd_dist.SharedFilePlyStream(dist, n_refs, per_round, path, torch.device("cpu"))
        for i, g in enumerate(mine):
            if counts[g]:
                st.push(i, torch.from_numpy(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        recs, gcounts = st.finish()               # this rank's own records, round by round, where they are (nothing concatenated)
        q.put((rank, sum(int(r.numel()) for r in recs), gcounts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs,world,per_round", [(7, 2, 2), (10, 3, 1), (5, 4, 3), (2, 3, 1)])
def test_shared_file_stream_writes_the_single_writer_file(n_refs, world, per_round, tmp_path):
    from lichtfeld_densification_plugin_amd.core.writers import StreamedPlyWriter
    path = os.path.join(str(tmp_path), "out", "shared.ply")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_file_worker, args=(r, world, port, n_refs, per_round, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs, seed=3)
    for rank, n_bytes, gcounts in results:
        np.testing.assert_array_equal(gcounts, counts)
        assert n_bytes == 15 * sum(counts[g] for g in lfd_dist.shard_references(n_refs, rank, world))      # every rank keeps its own shard
    ref = os.path.join(str(tmp_path), "single.ply")
    with StreamedPlyWriter(ref) as w:
        for g in range(n_refs):
            if counts[g]:
                w.append_packed(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).tobytes())
    assert open(path, "rb").read() == open(ref, "rb").read()


def _shared_file_failure_worker(rank, world, port, n_refs, path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs, seed=3)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        st = lfd_dist.SharedFilePlyStream(dist, n_refs, 2, path, torch.device("cpu"))
        if rank == 1:                                   # this rank's descriptor goes bad under it (disk error, file removed and the mount gone ...)
            os.close(st._fd)
            st._fd = os.open(os.devnull, os.O_RDONLY)   # pwrite on a read-only descriptor: EBADF
        for i, g in enumerate(mine):
            if counts[g]:
                st.push(i, torch.from_numpy(np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        try:
            st.finish()
            q.put((rank, None))
        except BaseException as exc:                    # noqa: BLE001
            q.put((rank, type(exc).__name__ + ": " + str(exc)))
    finally:
        dist.destroy_process_group()


def test_shared_file_stream_with_a_rank_that_cannot_write_raises_everywhere_and_leaves_an_empty_cloud(tmp_path):
    """one rank's writes fail: the rounds still match (nobody hangs), EVERY rank raises, and the header keeps `element vertex 0` - no reader
    takes the holes of an incomplete file for points"""
    n_refs, world = 9, 3
    path = os.path.join(str(tmp_path), "shared.ply")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_file_failure_worker, args=(r, world, port, n_refs, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "OSError" in results[1] or "Errno" in results[1], results
    assert "another rank failed" in results[0] and "another rank failed" in results[2], results
    head = open(path, "rb").read(200)
    assert int(head.split(b"element vertex")[1].split(b"\n")[0]) == 0          # (the count field is fixed-width, blank-padded)


# ---- records that travel from where they are (views of one launch buffer), placement into a caller's buffer, replication plan ----------------------
def _view_worker(rank, world, port, n_refs, n_rep, per_round, record, q):
    """What bench.py's sharded leg does, on host tensors: the last n_rep references are computed by every rank (written in place behind the
    exchanged part of ONE cloud buffer), the others are sharded; every round's records are consecutive views of one roomy buffer handed over
    with push_many (nothing concatenated, nothing padded); finish(place=...) puts the ordered records in front of the replicated ones."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs, seed=5)
        cols, dt = (7, torch.float32) if record == "f32" else (15, torch.uint8)

        def rec_of(g):
            return pts[g] if record == "f32" else np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15])
        mine, n_sh = lfd_dist.split_replicated(n_refs, n_rep, rank, world)
        sh = [g for g in mine if g < n_sh]
        rep = [g for g in mine if g >= n_sh]
        cap_sh = 40 * n_sh
        cloud = torch.zeros((cap_sh + 40 * len(rep), cols), dtype=dt)
        n_rep_rows = 0
        for g in rep:                                                     # "the kernel writes the replicated part in place"
            r = torch.from_numpy(rec_of(g))
            cloud[cap_sh + n_rep_rows:cap_sh + n_rep_rows + r.shape[0]] = r
            n_rep_rows += r.shape[0]
        ex = lfd_dist.OverlappedExchange(dist, n_sh, per_round, torch.device("cpu"), form="all_gather", record=record)
        views = 0
        for c0 in range(0, len(sh), per_round):
            chunk = sh[c0:c0 + per_round]
            buf = torch.full((40 * per_round + 7, cols), 77, dtype=dt)    # a launch's buffer: room for H * W records per reference, stale bytes behind
            cs = [int(counts[g]) for g in chunk]
            lo = 0
            for g in chunk:
                buf[lo:lo + counts[g]] = torch.from_numpy(rec_of(g))
                lo += counts[g]
            ex.push_many(c0, buf, cs)
        placed = {}

        def place(n_rows):
            placed["n"] = n_rows
            return cloud[cap_sh - n_rows:cap_sh]
        recs, gcounts = ex.finish(place=place)
        for st in ex._rounds:                                             # every payload is a view of its launch buffer, never a copy
            if st["n_local"] and st["padded"] is not None:
                views += int(st["padded"].untyped_storage().data_ptr() == st["payload"].untyped_storage().data_ptr())
        out = cloud[cap_sh - placed.get("n", 0):cap_sh + n_rep_rows]
        q.put((rank, out.numpy().copy(), gcounts, views, sum(1 for st in ex._rounds if st["n_local"] and st["padded"] is not None)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs,n_rep,world,per_round,record", [(9, 3, 2, 2, "ply"), (9, 0, 2, 2, "f32"), (11, 5, 3, 1, "ply"), (6, 6, 2, 2, "f32"), (8, 2, 4, 1, "f32")])
def test_views_placement_and_replication_give_the_single_process_cloud(n_refs, n_rep, world, per_round, record):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_view_worker, args=(r, world, port, n_refs, n_rep, per_round, record, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs, seed=5)
    full = np.concatenate(pts, 0)
    expect = full if record == "f32" else np.ascontiguousarray(full.view(np.uint8).reshape(-1, 28)[:, :15])
    for rank, cloud, gcounts, views, sent in results:
        np.testing.assert_array_equal(gcounts, counts[:n_refs - n_rep])
        np.testing.assert_array_equal(cloud, expect)                      # sharded part | replicated part: the 1-rank sequence, contiguous
        assert views == sent                                              # no round was copied into a padded buffer


def test_replication_plan():
    P = lfd_dist.plan_replication
    # the bare hot path on config 4 (one reference 5 us of kernel, 3.6 MB of survivors, 122 GB/s per peer): moving costs more than recomputing
    p2, p8 = P(56, 2, 0.0054, 3.57e6, launch_ms=0.01), P(56, 8, 0.0054, 3.57e6, launch_ms=0.01)
    assert p2["n_replicated"] > p8["n_replicated"] > 0 and p2["n_sharded"] + p2["n_replicated"] == 56
    for p in (p2, p8):
        assert p["step_ms"] <= p["pure_sharding_ms"] and p["step_ms"] <= p["single_rank_ms"] * 1.000001      # never worse than either extreme
    # a faster link shifts the plan towards sharding, a slower one towards replication; no link at all -> every rank computes everything
    assert P(56, 8, 0.0054, 3.57e6, link_gbps=1000.0)["n_sharded"] > p8["n_sharded"] > P(56, 8, 0.0054, 3.57e6, link_gbps=10.0)["n_sharded"]
    assert P(56, 4, 0.0054, 3.57e6, link_gbps=1e-3)["n_sharded"] == 0
    # with a matcher in the loop (8 pairs x 30 ms per reference) nothing is worth recomputing
    assert P(56, 8, 240.0, 3.57e6)["n_replicated"] == 0 and P(148, 2, 90.0, 3.4e6)["n_replicated"] == 0
    # one rank: nothing to exchange
    p1 = P(56, 1, 0.0054, 3.57e6)
    assert p1["n_replicated"] == 0 and p1["step_ms"] == p1["single_rank_ms"]
    assert lfd_dist.split_replicated(10, 4, 1, 3) == ([1, 4, 6, 7, 8, 9], 6) and lfd_dist.split_replicated(10, 4, 1, 3, replicas_here=False) == ([1, 4], 6)
    assert lfd_dist.split_replicated(5, 9, 0, 2) == ([0, 1, 2, 3, 4], 0)


def _eager_worker(rank, world, port, n_refs, per_round, form, record, use_dest, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, pts = _make_points(n_refs, seed=9)
        cols, dt = (7, torch.float32) if record == "f32" else (15, torch.uint8)
        mine = lfd_dist.shard_references(n_refs, rank, world)
        dest = torch.full((int(counts.sum()) + 3, cols), 9, dtype=dt) if use_dest else None
        ex = lfd_dist.OverlappedExchange(dist, n_refs, per_round, torch.device("cpu"), form=form, record=record, eager=True, dest=dest)
        for i, g in enumerate(mine):
            if counts[g]:
                ex.push(i, torch.from_numpy(pts[g] if record == "f32" else np.ascontiguousarray(pts[g].view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)))
        recs, gcounts = ex.finish()
        shares = use_dest and (form == "all_gather" or rank == 0) and recs.numel() > 0 and recs.untyped_storage().data_ptr() == dest.untyped_storage().data_ptr()
        q.put((rank, recs.numpy().copy(), gcounts, bool(shares)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs,world,per_round,form,record,use_dest", [
    (7, 2, 1, "all_gather", "f32", True), (7, 2, 2, "gather_to_root", "ply", True), (10, 3, 2, "all_gather", "ply", False), (5, 4, 1, "gather_to_root", "f32", False),
    (9, 2, 16, "all_gather", "ply", True), (2, 3, 1, "all_gather", "f32", True)])
def test_eager_rounds_and_early_placement_are_the_same_sequence(n_refs, world, per_round, form, record, use_dest):
    """eager=True (a round's records leave as soon as it is complete) and dest= (ordered records placed from row 0 of the caller's buffer as the
    rounds complete): the 1-rank sequence, in the caller's buffer where one was given"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eager_worker, args=(r, world, port, n_refs, per_round, form, record, use_dest, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs, seed=9)

    def as_sent(parts):
        if not parts:
            return np.zeros((0, 7), np.float32) if record == "f32" else np.zeros((0,), np.uint8)
        full = np.concatenate(parts, 0)
        return full if record == "f32" else np.ascontiguousarray(full.view(np.uint8).reshape(-1, 28)[:, :15]).reshape(-1)
    for rank, recs, gcounts, shares in results:
        np.testing.assert_array_equal(gcounts, counts)
        have = form == "all_gather" or rank == 0
        np.testing.assert_array_equal(recs, as_sent(pts) if have else as_sent([pts[g] for g in lfd_dist.shard_references(n_refs, rank, world)]))
        if use_dest and have and int(counts.sum()):
            assert shares                                  # the result IS the caller's buffer, not a copy
