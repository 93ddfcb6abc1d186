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
    """Records the calls; a group's result: reference with uid u yields u % 5 + 1 points whose x coordinate is u (status 3 for uids in `refused`).
    The random stream is a counter: reference u draws u + 1 numbers, a point's y coordinate is where the stream stood when its reference
    began - what a wrong stream shows up in.  Failures of the device code, as the kernels report them: `broken` - the chain's bounded waits
    expire AT that reference (status 5 from there on, nothing committed: the stream stays where the call found it although the references
    before it have drawn); `inexact` - that reference is refused with status 4 and draws nothing (upstream would have drawn)."""

    def __init__(self, normaliser=True, refused=(), unbatchable=(), broken=(), inexact=(), launch_fails=(), sums_fail=(), collect_fails=()):
        self.clock, self.calls, self.normaliser = _Clock(), [], normaliser
        self.refused, self.unbatchable = set(refused), set(unbatchable)
        self.broken, self.inexact, self.launch_fails, self.sums_fail, self.collect_fails = set(broken), set(inexact), set(launch_fails), set(sums_fail), set(collect_fails)
        self.pos, self.saved = 0, {}

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
        if any(r in self.sums_fail for r in slot["batch"].refs):
            raise RuntimeError("pinned copy failed")
        return [float(r) + 0.5 for r in slot["batch"].refs]

    def checkpoint_rng(self, place):
        self.calls.append(("checkpoint", place))
        self.saved[place] = self.pos

    def rollback_rng(self, place):
        self.calls.append(("rollback", place))
        self.pos = self.saved[place]

    def launch_sampled_chain(self, batch, sums):
        self.calls.append(("launch", tuple(batch.refs), None if sums is None else tuple(sums)))
        # (the stand-in computes at launch what the device computes in stream order)
        pos, offs, rows, st, broke = self.pos, [0], [], [], False
        for u in batch.refs:
            broke = broke or u in self.broken
            n = 0
            if broke:
                st.append(5)
            elif u in self.inexact:
                st.append(4)
            elif u in self.refused:
                st.append(3)
            else:
                st.append(0)
                n = u % 5 + 1
                rows += [[float(u), float(pos), 0.0]] * n
                pos += u + 1
            offs.append(offs[-1] + n)
        if not broke:
            self.pos = pos                    # a broken chain commits nothing
        if any(u in self.launch_fails for u in batch.refs):
            self.pos += 1000                  # enqueued in part: the stream has moved, the call raises
            self.launch_fails -= set(batch.refs)
            raise RuntimeError("out of memory")
        xyz = torch.tensor(rows, dtype=torch.float32).reshape(-1, 3)
        return batch, hb.TriangulationOutput(xyz=xyz, rgb=xyz.clone(), err=xyz[:, 0].clone(), cell=None, slot=None, ref_offsets=np.asarray(offs, np.int64),
                                             seg_counts=None, seg_order=None, sel_status=np.asarray(st, np.int32))

    def finish_sampled(self, handle, check_selection=True):
        batch, res = handle
        self.calls.append(("collect", tuple(batch.refs)))
        if any(u in self.collect_fails for u in batch.refs):
            raise RuntimeError("launch status 3")
        return res

    def sampled(self, ref, axes, rng, dseed, need_best=False):
        self.calls.append(("single", ref))
        if ref in self.refused:
            raise ValueError("Fewer non-zero entries in p than size")
        n = ref % 5 + 1
        xyz = torch.tensor([[float(ref), float(self.pos), 0.0]] * n, dtype=torch.float32)
        self.pos += ref + 1                   # (a reference the device stage refuses for inexactness goes through the host stage here and draws)
        return hb.TriangulationOutput(xyz=xyz, rgb=xyz.clone(), err=xyz[:, 0].clone(), cell=torch.zeros(n, dtype=torch.int32), slot=torch.zeros(n, dtype=torch.uint8),
                                      ref_offsets=np.asarray([0, n], np.int64), seg_counts=None, seg_order=None), None

    def debug_matches(self, *a):
        return np.zeros((0, 4), np.float32)


class _Out:
    def __init__(self):
        self.emitted = []

    def emit(self, em, hot):
        self.emitted.append((em.packed.ref_uid, int(em.points[0].shape[0]), float(em.points[0][0, 0]), em.dbg is not None, float(em.points[0][0, 1])))


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


def _plain_stream(uids, refused=()):
    """where the stream stands when each reference begins in the one-reference-at-a-time schedule (a refused reference draws nothing)"""
    pos, out = 0, {}
    for u in uids:
        out[u] = float(pos)
        if u not in refused:
            pos += u + 1
    return out, pos


def _run(hot, n_refs, group):
    loop, out = _loop(hot, group)
    for u in range(n_refs):
        loop.submit(_m(u))
    loop.drain()
    return loop, out


def test_a_checkpoint_of_the_stream_precedes_every_grouped_call():
    hot = _Hot()
    _run(hot, 9, 3)
    kinds = [c[0] for c in hot.calls if c[0] in ("checkpoint", "launch")]
    assert kinds == ["checkpoint", "launch"] * 3 and not any(c[0] == "rollback" for c in hot.calls)
    assert len({c[1] for c in hot.calls if c[0] == "checkpoint"}) == 3                                  # places in rotation: two calls are in flight at most


def test_a_chain_that_breaks_is_void_with_everything_launched_behind_it_and_redone_on_the_restored_stream(caplog):
    # Group (3, 4, 5) breaks at reference 4: reference 3 has drawn, nothing was committed, and group (6, 7, 8) - launched before the status of
    # (3, 4, 5) was read - started from the uncommitted stream.  Every reference must end up with the stream position of the plain schedule.
    hot = _Hot(broken={4})
    hot_ok = _Hot()
    loop, out = _run(hot, 11, 3)
    _loop_ok, out_ok = _run(hot_ok, 11, 3)
    expect, end = _plain_stream(range(11))
    assert [e[0] for e in out.emitted] == list(range(11)) and [e[4] for e in out.emitted] == [expect[u] for u in range(11)]
    assert out.emitted == out_ok.emitted and hot.pos == hot_ok.pos == end
    singles = [c[1] for c in hot.calls if c[0] == "single"]
    assert singles == [3, 4, 5, 6, 7, 8]                                                               # the void calls' references, one by one, in order
    kinds = [(c[0], c[1]) for c in hot.calls]
    assert kinds.index(("collect", (6, 7, 8))) < kinds.index(("single", 3))                            # the later call is waited for (and dropped) first
    assert [c for c in hot.calls if c[0] == "rollback"] == [("rollback", 1)]                           # to the checkpoint taken before (3, 4, 5)
    assert [c[1] for c in hot.calls if c[0] == "launch"] == [(0, 1, 2), (3, 4, 5), (6, 7, 8), (9, 10)]  # the group behind the recovery is launched as ever
    assert any("redoing refs [3, 4, 5, 6, 7, 8] one by one" in r.getMessage() for r in caplog.records)


def test_a_reference_refused_for_inexactness_voids_its_call_because_upstream_would_have_drawn():
    hot = _Hot(inexact={1})
    _loop_, out = _run(hot, 6, 3)
    expect, end = _plain_stream(range(6))
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(6)] and hot.pos == end
    assert [c[1] for c in hot.calls if c[0] == "single"] == [0, 1, 2, 3, 4, 5]                          # (3, 4, 5) had been launched behind it


def test_a_grouped_call_that_raises_half_way_is_redone_from_its_checkpoint():
    hot = _Hot(launch_fails={4})
    _loop_, out = _run(hot, 8, 3)
    expect, end = _plain_stream(range(8))
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(8)] and hot.pos == end
    kinds = [(c[0], c[1]) for c in hot.calls]
    assert [c[1] for c in hot.calls if c[0] == "single"] == [3, 4, 5]
    assert kinds.index(("collect", (0, 1, 2))) < kinds.index(("rollback", 1)) < kinds.index(("single", 3))   # the call before it is emitted first
    assert ("launch", (6, 7)) in kinds


def test_sums_that_cannot_be_taken_cost_no_rollback_and_no_reference():
    hot = _Hot(sums_fail={3})
    _loop_, out = _run(hot, 8, 3)
    expect, end = _plain_stream(range(8))
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(8)] and hot.pos == end
    assert not any(c[0] == "rollback" for c in hot.calls) and [c[1] for c in hot.calls if c[0] == "single"] == [3, 4, 5]


def test_a_call_whose_collection_fails_is_void_too():
    hot = _Hot(collect_fails={0})
    _loop_, out = _run(hot, 7, 3)
    expect, end = _plain_stream(range(7))
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(7)] and hot.pos == end


def test_two_failures_in_a_row_use_only_the_first_checkpoint():
    # (0, 1, 2) breaks; (3, 4, 5), launched behind it, raises half way: the recovery of the first call redoes both, and the second call's checkpoint -
    # taken on a void stream - is never rolled back to
    hot = _Hot(broken={1}, launch_fails={4})
    _loop_, out = _run(hot, 8, 3)
    expect, end = _plain_stream(range(8))
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(8)] and hot.pos == end
    assert [c for c in hot.calls if c[0] == "rollback"] == [("rollback", 0)]


def test_upstreams_own_refusal_stays_one_references_error_inside_a_recovered_call(caplog):
    hot = _Hot(broken={5}, refused={4})
    _loop_, out = _run(hot, 7, 3)
    expect, end = _plain_stream(range(7), refused={4})
    assert [(e[0], e[4]) for e in out.emitted] == [(u, expect[u]) for u in range(7) if u != 4] and hot.pos == end
    assert any("Triangulation error for ref 4: Fewer non-zero entries in p than size" in r.getMessage() for r in caplog.records)


def test_an_automatic_group_is_bounded_by_the_bytes_of_its_buffers():
    from lichtfeld_densification_plugin_amd.core.strategies import AUTO_DEVICE_BYTES, AUTO_PINNED_BYTES, bounded_group
    assert bounded_group(16, 512 * 512, 33, 0) == 16 and bounded_group(16, 512 * 512, 15, 15) == 16 and bounded_group(16, 512 * 512, 8, 4) == 16
    assert bounded_group(16, 1280 * 1280, 15, 15) == AUTO_PINNED_BYTES // (1280 * 1280 * 15) == 10      # dense streamer at `precise`
    assert bounded_group(16, 1280 * 1280, 33, 0) == 16 and bounded_group(16, 4096 * 4096, 33, 0) == AUTO_DEVICE_BYTES // (4096 * 4096 * 33) == 1
    assert bounded_group(16, 1 << 40, 33, 0) == 1                                                       # never less than one
    hot = _Hot(normaliser=False)
    cfg = lfd.DensePipelineConfig(output_path="a.ply", refs_per_launch=16)
    out = _Out()
    loop = SampledLoop(hot, out, cfg, per_ref_rng=False, auto_group=True)
    for u in range(8):
        loop.submit(_m(u, size=4096))                 # 16 Mcells: 64 MiB of pinned weights per reference - four fit the landing area
    loop.drain()
    assert [c[1] for c in hot.calls if c[0] == "launch"] == [(0, 1, 2, 3), (4, 5, 6, 7)]
