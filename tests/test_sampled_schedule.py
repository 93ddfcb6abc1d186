"""The sampled loop's grouped schedule on upstream's one RNG stream (core/strategies.py::SampledLoop, lfd_triangulate_sampled_chain) with a stand-in for
the hot path: what is launched in which order, what is emitted in which order, and what happens to a reference - or a group - that fails.  No GPU."""
import types

import numpy as np
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from lichtfeld_densification_plugin_amd.core.strategies import Matched, SampledLoop


class _Clock:
    serialising = False


class _Hot:
    """Records the calls; a group's result: reference with uid u yields u % 5 + 1 points whose x coordinate is u (status 3 for uids in `refused`)."""

    def __init__(self, normaliser=True, refused=(), unbatchable=()):
        self.clock, self.calls, self.normaliser = _Clock(), [], normaliser
        self.refused, self.unbatchable = set(refused), set(unbatchable)

    def can_launch_ahead(self, *a):
        return False

    def can_pipeline_normaliser(self, *a):
        return False

    def can_chain(self, need_best, H, W):
        return not need_best

    def chain_uses_upstream_normaliser(self):
        return self.normaliser

    def prepare_chain(self, refs, axes):
        if any(r in self.unbatchable for r in refs):
            raise ValueError("cert: expected torch.float32 (64, 64)")
        self.calls.append(("prepare", tuple(refs)))
        return types.SimpleNamespace(refs=list(refs), n_refs=len(refs))

    def begin_chain_normalisers(self, batch):
        self.calls.append(("begin", tuple(batch.refs)))
        return {"batch": batch}

    def finish_chain_normalisers(self, slot):
        self.calls.append(("sums", tuple(slot["batch"].refs)))
        return [float(r) + 0.5 for r in slot["batch"].refs]

    def launch_sampled_chain(self, batch, sums):
        self.calls.append(("launch", tuple(batch.refs), None if sums is None else tuple(sums)))
        return batch, None

    def finish_sampled(self, handle, check_selection=True):
        batch, _ = handle
        self.calls.append(("collect", tuple(batch.refs)))
        offs, rows, st = [0], [], []
        for u in batch.refs:
            n = 0 if u in self.refused else u % 5 + 1
            rows += [[float(u), 0.0, 0.0]] * n
            offs.append(offs[-1] + n)
            st.append(3 if u in self.refused else 0)
        xyz = torch.tensor(rows, dtype=torch.float32).reshape(-1, 3)
        return hb.TriangulationOutput(xyz=xyz, rgb=xyz.clone(), err=xyz[:, 0].clone(), cell=None, slot=None, ref_offsets=np.asarray(offs, np.int64),
                                      seg_counts=None, seg_order=None, sel_status=np.asarray(st, np.int32))

    def sampled(self, ref, axes, rng, dseed, need_best=False):
        self.calls.append(("single", ref))
        n = ref % 5 + 1
        xyz = torch.full((n, 3), float(ref))
        return hb.TriangulationOutput(xyz=xyz, rgb=xyz.clone(), err=xyz[:, 0].clone(), cell=torch.zeros(n, dtype=torch.int32), slot=torch.zeros(n, dtype=torch.uint8),
                                      ref_offsets=np.asarray([0, n], np.int64), seg_counts=None, seg_order=None), None

    def debug_matches(self, *a):
        return np.zeros((0, 4), np.float32)


class _Out:
    def __init__(self):
        self.emitted = []

    def emit(self, em, hot):
        self.emitted.append((em.packed.ref_uid, int(em.points[0].shape[0]), float(em.points[0][0, 0]), em.dbg is not None))


class _Ref(int):
    """the stand-in for hb.ReferenceInputs: the reference's number, with the one attribute the debug preview reads"""
    cert = (None, None)


def _m(uid, want_debug=False, size=64):
    packed = types.SimpleNamespace(ref_uid=uid)
    return Matched(local_i=uid, packed=packed, ref=_Ref(uid), axes=None, H=size, W=size, first_pair=0, want_debug=want_debug)


def _loop(hot, n):
    cfg = lfd.DensePipelineConfig(output_path="a.ply", refs_per_launch=n)
    out = _Out()
    return SampledLoop(hot, out, cfg, per_ref_rng=False), out


def test_groups_are_launched_one_behind_their_weight_maps_and_emitted_in_order():
    hot = _Hot()
    loop, out = _loop(hot, 3)
    for u in range(8):
        loop.submit(_m(u))
    loop.drain()
    assert [e[0] for e in out.emitted] == list(range(8)) and all(e[1] == e[0] % 5 + 1 and e[2] == float(e[0]) for e in out.emitted)
    kinds = [(c[0], c[1]) for c in hot.calls]
    # group (3,4,5)'s weight maps start their way before group (0,1,2) gets its sums and its fused call; a group is collected after the next launch
    assert kinds.index(("begin", (3, 4, 5))) < kinds.index(("sums", (0, 1, 2))) < kinds.index(("launch", (0, 1, 2)))
    assert kinds.index(("launch", (3, 4, 5))) < kinds.index(("collect", (0, 1, 2)))
    assert kinds.index(("launch", (6, 7))) < kinds.index(("collect", (3, 4, 5)))                       # the last, short group at drain()
    launches = [c for c in hot.calls if c[0] == "launch"]
    assert [c[1] for c in launches] == [(0, 1, 2), (3, 4, 5), (6, 7)] and launches[0][2] == (0.5, 1.5, 2.5)      # upstream's sums, reference by reference


def test_device_sums_need_no_look_ahead():
    hot = _Hot(normaliser=False)
    loop, out = _loop(hot, 4)
    for u in range(4):
        loop.submit(_m(u))
    assert [c for c in hot.calls if c[0] == "launch"] == [("launch", (0, 1, 2, 3), None)]              # launched as soon as the group is full
    loop.drain()
    assert [e[0] for e in out.emitted] == [0, 1, 2, 3] and not any(c[0] in ("begin", "sums") for c in hot.calls)


def test_a_refused_reference_is_its_own_error_and_a_debug_reference_runs_alone_in_order(caplog):
    hot = _Hot(refused={4})
    loop, out = _loop(hot, 3)
    for u in range(7):
        loop.submit(_m(u, want_debug=(u == 2)))
    loop.drain()
    assert [e[0] for e in out.emitted] == [0, 1, 2, 3, 5, 6]                                           # 4 drew nothing and emits nothing; order kept
    assert [e[3] for e in out.emitted] == [False, False, True, False, False, False]                    # the debug reference carries its preview
    singles = [c for c in hot.calls if c[0] == "single"]
    launches = [c[1] for c in hot.calls if c[0] == "launch"]
    assert singles == [("single", 2)] and launches == [(0, 1), (3, 4, 5), (6,)]                        # the pending group goes out before the lone reference
    kinds = [(c[0], c[1]) for c in hot.calls]
    assert kinds.index(("collect", (0, 1))) < kinds.index(("single", 2)) < kinds.index(("launch", (3, 4, 5)))
    assert any("Triangulation error for ref 4: Fewer non-zero entries in p than size" in r.getMessage() for r in caplog.records)    # upstream's ValueError text


def test_a_group_that_cannot_be_batched_is_redone_one_by_one_behind_the_groups_before_it():
    hot = _Hot(unbatchable={4})
    loop, out = _loop(hot, 3)
    for u in range(6):
        loop.submit(_m(u))
    loop.drain()
    assert [e[0] for e in out.emitted] == [0, 1, 2, 3, 4, 5]
    kinds = [(c[0], c[1]) for c in hot.calls]
    assert [c for c in kinds if c[0] == "single"] == [("single", 3), ("single", 4), ("single", 5)]
    assert kinds.index(("collect", (0, 1, 2))) < kinds.index(("single", 3))                            # nothing of the failed group had drawn: order kept


def test_a_change_of_grid_closes_the_group():
    hot = _Hot(normaliser=False)
    loop, out = _loop(hot, 4)
    loop.submit(_m(0)); loop.submit(_m(1)); loop.submit(_m(2, size=48)); loop.submit(_m(3, size=48))
    loop.drain()
    assert [c[1] for c in hot.calls if c[0] == "launch"] == [(0, 1), (2, 3)] and [e[0] for e in out.emitted] == [0, 1, 2, 3]
