"""N4 pair scheduler: schedule arithmetic, last-use eviction, and the matcher mirror computing every camera's backbone features
once per run when it is handed camera keys (a stub RoMaV2 counts the backbone passes; no weights needed)."""
import sys
import types

import numpy as np
import pytest
import torch

from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import selection
from lichtfeld_densification_plugin_amd.core.scheduler import FeatureCache, PairSchedule


def _garden(frac, k):
    cams = synthetic.ring_cameras(185, seed=0)
    flat = np.stack([c.flat_pose() for c in cams])
    refs = selection.select_cameras_kcenters(flat, round(frac * 185))
    nn = selection.nearest_neighbors(flat, k)
    return cams, refs, nn


@pytest.mark.parametrize("frac,k,pairs", [(0.8, 3, 444), (0.75, 4, 556), (0.3, 8, 448)])
def test_schedule_counts_on_the_garden_ring(frac, k, pairs):
    """GUI defaults 148 x 3 = 444 pairs, CLI 139 x 4 = 556, config 4: 56 x 8 = 448; upstream runs one backbone pass
    per reference and one per pair, the shared schedule one per camera."""
    cams, refs, nn = _garden(frac, k)
    s = PairSchedule(refs, nn, [c.uid for c in cams], k)
    assert s.n_pairs == pairs and len(s.items) == len(refs)
    assert s.n_backbone_forwards_upstream == len(refs) + pairs
    assert s.n_backbone_forwards_shared == len(s.uses) <= 185
    assert [it.position for it in s.items] == list(range(len(refs)))          # upstream's order
    # sharding: the union of the ranks' schedules is the whole schedule
    parts = [PairSchedule(refs, nn, [c.uid for c in cams], k, positions=range(r, len(refs), 4)) for r in range(4)]
    assert sum(p.n_pairs for p in parts) == pairs


def test_feature_cache_computes_each_camera_once_and_evicts_at_last_use():
    cams, refs, nn = _garden(0.3, 8)
    s = PairSchedule(refs, nn, [c.uid for c in cams], 8)
    cache = FeatureCache(s.last_use)
    computed = []
    for step, it in enumerate(s.items):
        for cam in [it.ref_index] + it.nbr_indices:
            val = cache.get_or_compute(cam, lambda cam=cam: computed.append(cam) or ("feat", cam))
            assert val == ("feat", cam)
        cache.advance(step)
        assert all(s.last_use[c] > step for c in cache.keys())                 # nothing dead is kept
    assert sorted(computed) == sorted(s.uses) and len(computed) == s.n_backbone_forwards_shared
    assert cache.misses == len(computed) and cache.hits == s.n_backbone_forwards_upstream - len(computed)
    assert cache.peak <= s.peak_resident() + 9 and len(cache) == 0


class _StubDescriptor(torch.nn.Module):
    """Stands for romav2's Descriptor (DINOv3): an nn.Module with a parameter, counting its forward passes."""

    def __init__(self):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.ones(1), requires_grad=False)
        self.calls = 0

    def forward(self, img):
        self.calls += 1
        return [img.mean(dim=(2, 3)) * self.scale]


class _StubRoMa(torch.nn.Module):
    """The slice of RoMaV2's interface core/matcher.py touches.  Like the real model (RoMaV2/src/romav2/romav2.py:68,101,177) it
    is an nn.Module whose backbone ``f`` is a REGISTERED CHILD MODULE and whose own code calls ``self.f(img_B_lr)``: assigning
    anything but an nn.Module to ``model.f`` raises TypeError, which a plain-class stub would not show."""

    class Cfg:
        def __init__(self, compile=False):
            pass

    def __init__(self, cfg=None):
        super().__init__()
        self.f = _StubDescriptor()
        self.H_lr = self.W_lr = 32
        self.H_hr = self.W_hr = None
        self.bidirectional = False

    @property
    def backbone_calls(self):
        return self.f.calls

    def apply_setting(self, s):
        pass

    def _load_image(self, im):
        if isinstance(im, torch.Tensor):
            return im.float() / 255.0 if im.dtype == torch.uint8 else im
        return torch.from_numpy(np.array(im)).permute(2, 0, 1).unsqueeze(0).float() / 255.0

    def _forward_from_features(self, f_list_A, img_A_lr, img_B_lr):
        f_b = self.f(img_B_lr)                       # romav2.py:177
        shift = (f_list_A[0] - f_b[0]).mean()
        warp = torch.zeros((1, self.H_lr, self.W_lr, 2), device=img_B_lr.device) + shift
        return {"warp_AB": warp, "overlap_AB": torch.full((1, self.H_lr, self.W_lr, 1), 0.5, device=img_B_lr.device)}

    def match_from_features(self, f_list_A, img_A_lr, imB, img_A_hr=None):
        # RoMaV2._resize_match_image (romav2.py:323-333): the same bicubic antialiased resize the mirror applies to the reference
        img_b = torch.nn.functional.interpolate(self._load_image(imB), size=(self.H_lr, self.W_lr), mode="bicubic", align_corners=False, antialias=True)
        return self._forward_from_features(f_list_A, img_A_lr, img_b)


def test_the_stub_refuses_what_the_real_model_refuses():
    """nn.Module.__setattr__ only takes a Module (or None) for a registered child: the round-2 hook (a lambda) dies here."""
    m = _StubRoMa()
    with pytest.raises(TypeError):
        m.f = lambda img: [img]
    assert "f.scale" in m.state_dict()


def test_matcher_mirror_shares_backbone_features_between_references(monkeypatch):
    from PIL import Image
    stub = types.ModuleType("romav2")
    stub.RoMaV2 = _StubRoMa
    monkeypatch.setitem(sys.modules, "romav2", stub)
    from lichtfeld_densification_plugin_amd.core import matcher as mm
    rs = np.random.RandomState(0)
    images = {i: Image.fromarray(rs.randint(0, 256, (40, 48, 3)).astype(np.uint8)) for i in range(6)}
    work = [(0, [1, 2]), (1, [0, 2]), (2, [1, 3]), (4, [3, 5])]
    uids = list(range(6))
    nn = {0: [1, 2], 1: [0, 2], 2: [1, 3], 4: [3, 5], 3: [], 5: []}
    sched = PairSchedule([w[0] for w in work], nn, uids, 2)

    def run(keys):
        m = mm.RomaMatcher(device="cpu", setting="turbo")
        cache = FeatureCache(sched.last_use)
        if keys:
            m.set_feature_cache(cache)
        outs = []
        for step, (r, nbrs) in enumerate(work):
            outs.append(m.match_grids_batch(images[r], [images[n] for n in nbrs], **({"keys": (r, nbrs)} if keys else {})))
            cache.advance(step)
        return m.model.backbone_calls, outs

    plain_calls, plain = run(False)
    shared_calls, shared = run(True)
    assert plain_calls == sched.n_backbone_forwards_upstream == 12 and shared_calls == sched.n_backbone_forwards_shared == 6
    for a, b in zip(plain, shared):
        for (wa, ca), (wb, cb) in zip(a, b):
            assert torch.equal(wa, wb) and torch.equal(ca, cb)                  # same maps either way


def test_keyed_call_leaves_the_model_as_it_found_it(monkeypatch):
    """The wrapper is installed for the duration of a call only: afterwards (and after an exception inside the model) ``model.f``
    is the original child again and the state dict has the original keys; features of another input size are never served."""
    from PIL import Image
    stub = types.ModuleType("romav2")
    stub.RoMaV2 = _StubRoMa
    monkeypatch.setitem(sys.modules, "romav2", stub)
    from lichtfeld_densification_plugin_amd.core import matcher as mm
    rs = np.random.RandomState(1)
    ims = [Image.fromarray(rs.randint(0, 256, (40, 48, 3)).astype(np.uint8)) for _ in range(3)]
    m = mm.RomaMatcher(device="cpu", setting="turbo")
    original = m.model.f
    keys_before = sorted(m.model.state_dict())
    cache = FeatureCache({0: 5, 1: 5, 2: 5})
    m.set_feature_cache(cache)
    m.match_grids_batch(ims[0], [ims[1], ims[2]], keys=(0, [1, 2]))
    assert m.model.f is original and sorted(m.model.state_dict()) == keys_before and original.calls == 3

    def boom(*a, **k):
        raise RuntimeError("inside the model")
    monkeypatch.setattr(m.model, "match_from_features", boom)
    with pytest.raises(RuntimeError, match="inside the model"):
        m.match_grids_batch(ims[0], [ims[1]], keys=(0, [1]))
    assert m.model.f is original
    monkeypatch.undo()
    monkeypatch.setitem(sys.modules, "romav2", stub)
    # the same matcher at another resolution: camera 0's 32x32 features must not answer a 48x48 request
    m.model.H_lr = m.model.W_lr = 48
    calls = original.calls
    m.match_grids_batch(ims[0], [ims[1]], keys=(0, [1]))
    assert original.calls == calls + 2
    with pytest.raises(ValueError):
        m.match_grids_batch(ims[0], [ims[1], ims[2]], keys=(0, [1]))
    # sharing switched off again: keys are ignored, every call goes to the backbone
    m.set_feature_cache(None)
    calls = original.calls
    m.match_grids_batch(ims[0], [ims[1]], keys=(0, [1]))
    assert original.calls == calls + 2 and m.model.f is original


class _BatchStubRoMa(_StubRoMa):
    """... with RoMaV2's batched entry points (romav2.py:323-372: _resize_match_image and _match_core work on a batch)."""

    def _resize_match_image(self, img):
        return torch.nn.functional.interpolate(img, size=(self.H_lr, self.W_lr), mode="bicubic", align_corners=False, antialias=True), None

    def _match_core(self, f_list_A, img_A_lr, img_B_lr, img_A_hr=None, img_B_hr=None):
        f_b = self.f(img_B_lr)                                           # one backbone pass for the whole batch
        n = img_B_lr.shape[0]
        shift = (f_list_A[0] - f_b[0]).mean(dim=1).view(n, 1, 1, 1)      # per pair, the arithmetic of _forward_from_features
        warp = torch.zeros((n, self.H_lr, self.W_lr, 2)) + shift
        return {"warp_AB": warp, "overlap_AB": torch.full((n, self.H_lr, self.W_lr, 1), 0.5)}

    def _forward_from_features(self, f_list_A, img_A_lr, img_B_lr):
        return self._match_core(f_list_A, img_A_lr, img_B_lr)


@pytest.mark.parametrize("share", [False, True])
def test_several_pairs_per_forward_equal_one_pair_per_forward(monkeypatch, share):
    """N4, second half: ``pairs_per_forward`` = P stacks P neighbours of a reference into one model forward (upstream's loop runs one
    pair per forward, core/matcher.py:175-188; RoMaV2's own tests run batches of 8).  On a model whose arithmetic is per sample the maps
    are the same bits; the backbone is CALLED fewer times, and with shared features it still sees every camera exactly once."""
    from PIL import Image
    stub = types.ModuleType("romav2")
    stub.RoMaV2 = _BatchStubRoMa
    monkeypatch.setitem(sys.modules, "romav2", stub)
    from lichtfeld_densification_plugin_amd.core import matcher as mm
    rs = np.random.RandomState(3)
    images = {i: Image.fromarray(rs.randint(0, 256, (40, 48, 3)).astype(np.uint8)) for i in range(8)}
    work = [(0, [1, 2, 3, 4, 5]), (1, [0, 2, 6]), (2, [1, 3, 7, 5])]
    nn = {r: n for r, n in work}
    nn.update({i: [] for i in range(8) if i not in nn})
    sched = PairSchedule([w[0] for w in work], nn, list(range(8)), 5)

    def run(P):
        m = mm.RomaMatcher(device="cpu", setting="turbo", pairs_per_forward=P)
        images_seen = []
        inner_forward = m.model.f.forward
        m.model.f.forward = lambda img: (images_seen.append(int(img.shape[0])), inner_forward(img))[1]
        cache = FeatureCache(sched.last_use)
        if share:
            m.set_feature_cache(cache)
        outs = []
        for step, (r, nbrs) in enumerate(work):
            outs.append(m.match_grids_batch(images[r], [images[n] for n in nbrs], **({"keys": (r, nbrs)} if share else {})))
            cache.advance(step)
        return outs, images_seen

    single, seen1 = run(1)
    for P in (2, 3, 8):
        batched, seenP = run(P)
        assert sum(seenP) == sum(seen1) == (sched.n_backbone_forwards_shared if share else sched.n_backbone_forwards_upstream)
        assert len(seenP) < len(seen1) and max(seenP) <= P            # fewer, larger passes
        for a, b in zip(single, batched):
            assert len(a) == len(b)
            for (wa, ca), (wb, cb) in zip(a, b):
                assert torch.equal(wa, wb) and torch.equal(ca, cb)


class _RebindingStubRoMa(_StubRoMa):
    """... whose forward REBINDS the last level of the neighbour's feature list, as the real model does in bidirectional settings
    (``_compute_head_preds``: ``head_input = f_list; head_input[-1] = f_list[-1] + ...`` on the list ``self.f(img_B_lr)`` returned,
    RoMaV2/src/romav2/matcher.py:57-59,179-186)."""

    def _forward_from_features(self, f_list_A, img_A_lr, img_B_lr):
        f_b = self.f(img_B_lr)
        shift = (f_list_A[0] - f_b[0]).mean()
        f_b[-1] = f_b[-1] + 1000.0                    # what the head's input assembly does to the caller's list
        warp = torch.zeros((1, self.H_lr, self.W_lr, 2), device=img_B_lr.device) + shift
        return {"warp_AB": warp, "overlap_AB": torch.full((1, self.H_lr, self.W_lr, 1), 0.5, device=img_B_lr.device)}


def test_cached_features_survive_a_model_that_rebinds_its_feature_lists(monkeypatch):
    """Round 4, found by tests/golden/check_matcher_contract.py on the real class (`high`): a cached per-level list handed out as it is came
    back from a bidirectional forward with its last level replaced, and the camera's next use was matched on the wrong features.  The cache
    keeps an immutable snapshot and every call gets a fresh list."""
    from PIL import Image
    stub = types.ModuleType("romav2")
    stub.RoMaV2 = _RebindingStubRoMa
    monkeypatch.setitem(sys.modules, "romav2", stub)
    from lichtfeld_densification_plugin_amd.core import matcher as mm
    rs = np.random.RandomState(5)
    images = {i: Image.fromarray(rs.randint(0, 256, (40, 48, 3)).astype(np.uint8)) for i in range(3)}
    work = [(0, [1, 2]), (1, [0, 2]), (2, [0, 1])]

    def run(keys):
        m = mm.RomaMatcher(device="cpu", setting="turbo")
        if keys:
            m.set_feature_cache(FeatureCache({0: 9, 1: 9, 2: 9}))
        return [m.match_grids_batch(images[r], [images[n] for n in nbrs], **({"keys": (r, nbrs)} if keys else {})) for r, nbrs in work], m.model.backbone_calls

    plain, calls_plain = run(False)
    shared, calls_shared = run(True)
    assert calls_plain == 9 and calls_shared == 3
    for a, b in zip(plain, shared):
        for (wa, ca), (wb, cb) in zip(a, b):
            assert torch.equal(wa, wb) and torch.equal(ca, cb)
