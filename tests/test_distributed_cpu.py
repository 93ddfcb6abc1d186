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


def _worker(rank, world, port, n_refs, q):
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
        q.put((rank, gx.numpy(), gc.numpy(), ge.numpy(), gcounts, ax.numpy(), rc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_refs", [7, 2, 1])
def test_ordered_all_gather_world2(n_refs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_refs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, pts = _make_points(n_refs)
    full = np.concatenate(pts, 0)
    for rank, gx, gc, ge, gcounts, ax, rc in results:
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
