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
        assert all(s.last_use[c] > step for c in cache._store)                 # nothing dead is kept
    assert sorted(computed) == sorted(s.uses) and len(computed) == s.n_backbone_forwards_shared
    assert cache.misses == len(computed) and cache.hits == s.n_backbone_forwards_upstream - len(computed)
    assert cache.peak <= s.peak_resident() + 9 and len(cache) == 0


class _StubRoMa:
    """The slice of RoMaV2's interface core/matcher.py touches, with a counter on the backbone."""

    class Cfg:
        def __init__(self, compile=False):
            pass

    def __init__(self, cfg):
        self.H_lr = self.W_lr = 32
        self.H_hr = self.W_hr = None
        self.bidirectional = False
        self.backbone_calls = 0

    def apply_setting(self, s):
        pass

    def to(self, d):
        return self

    def eval(self):
        return self

    def _load_image(self, im):
        if isinstance(im, torch.Tensor):
            return im.float() / 255.0 if im.dtype == torch.uint8 else im
        return torch.from_numpy(np.array(im)).permute(2, 0, 1).unsqueeze(0).float() / 255.0

    def f(self, img):
        self.backbone_calls += 1
        return [img.mean(dim=(2, 3))]

    def match_from_features(self, f_list_A, img_A_lr, imB, img_A_hr=None):
        # RoMaV2._resize_match_image (romav2.py:323-333): the same bicubic antialiased resize the mirror applies to the reference
        img_b = torch.nn.functional.interpolate(self._load_image(imB), size=(self.H_lr, self.W_lr), mode="bicubic", align_corners=False, antialias=True)
        f_b = self.f(img_b)
        shift = (f_list_A[0] - f_b[0]).mean()
        warp = torch.zeros((1, self.H_lr, self.W_lr, 2)) + shift
        return {"warp_AB": warp, "overlap_AB": torch.full((1, self.H_lr, self.W_lr, 1), 0.5)}


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
