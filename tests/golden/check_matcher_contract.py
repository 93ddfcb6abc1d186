#!/usr/bin/env python3
"""Pin the RoMa producer shim (lichtfeld-densification-plugin_amd/core/matcher.py) against the REAL upstream classes - development
container only, like make_golden.py: /root/reference does not exist on the GPU box, and nothing of it is copied.

    python tests/golden/check_matcher_contract.py [--settings turbo fast high] [--write]

What runs: upstream's ``RomaMatcher`` (/root/reference/core/matcher.py:74-211) and this package's ``RomaMatcher`` each build ONE
``RoMaV2`` (/root/reference/RoMaV2/src/romav2/romav2.py:93-113) - the real class: real ``Matcher``, ``Refiners``, DPT head, local
correlation, bf16 autocast - from the same seed, so both hold the same (random) weights.  What the container lacks is stubbed with small
seeded modules of the right INTERFACE only:
    torch.hub.load(... "dinov3_vitl16")        -> a patch-embedding + two linear heads with ``get_intermediate_layers(x, n=[...])``
                                                  returning (B, H/16 * W/16, 1024) tokens (features.py:103-116)
    torchvision.models.vgg19_bn(weights=None)  -> the VGG19-BN ``features`` stack (conv3x3 / BatchNorm / ReLU / MaxPool by the public
                                                  configuration 'E'), random weights (features.py:166-174)
    torch.hub.load_state_dict_from_url         -> nothing to load (``load_state_dict`` is skipped: the seeded initialisation stays)
    lichtfeld                                  -> a logger
Then, for every setting, on seeded images of different sizes, this package's ``match_grids_batch`` must return upstream's
``(warp, cert)`` BIT FOR BIT (CPU) for: four-channel output (upstream's layout), two-channel output + ``reference_axes`` (the layout
the kernels consume in place), ``share_features`` off / on (the keyed descriptor swapped in for ``model.f``), and
``pairs_per_forward`` 1 / 3 - the last one is checked to tolerance and REPORTED when it is not bit-equal (batched GEMMs / bf16 convolutions
may round differently from single ones, which is why its default is 1).
``--write`` stores the hashes of upstream's outputs as tests/golden/g8_matcher_contract.json (a record of what was checked here;
the GPU box never sees upstream code)."""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE_ROOT = "/root/reference"
sys.path.insert(0, ROOT)


# ---- stand-ins for what the container lacks (interfaces only) ------------------------------------------------------------------------
class _TinyBackbone(nn.Module):
    """The interface of the DINOv3 hub model RoMaV2's Descriptor wraps: ``blocks`` (24) and ``get_intermediate_layers(x, n=[i, j])`` ->
    one (B, N, 1024) token tensor per requested layer.  Per-sample arithmetic only (a batch is processed image by image)."""

    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Identity() for _ in range(24)])
        self.embed = nn.Conv2d(3, 96, kernel_size=16, stride=16)
        self.heads = nn.ModuleList([nn.Linear(96, 1024) for _ in range(24)])

    def get_intermediate_layers(self, x, n):
        tok = self.embed(x).flatten(2).transpose(1, 2)            # (B, N, 96)
        return tuple(torch.tanh(self.heads[int(i)](tok)) for i in n)


def _vgg19_bn_features():
    cfg_e = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, c_in = [], 3
    for v in cfg_e:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c_in = v
    return nn.Sequential(*layers)


def install_stubs():
    lf = types.ModuleType("lichtfeld")

    class _Log:
        def info(self, m): pass
        def warn(self, m): pass
        def error(self, m): pass
        def debug(self, m): pass
    lf.log = _Log()
    sys.modules.setdefault("lichtfeld", lf)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")

        class _W:
            IMAGENET1K_V1 = None
        tvm.VGG19_BN_Weights = _W
        tvm.VGG19_Weights = _W

        def vgg19_bn(weights=None):
            m = nn.Module()
            m.features = _vgg19_bn_features()
            return m
        tvm.vgg19_bn = vgg19_bn
        tvm.vgg19 = vgg19_bn
        tv.models = tvm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
    torch.hub.load = lambda *a, **kw: _TinyBackbone()
    torch.hub.load_state_dict_from_url = lambda *a, **kw: {}
    src = os.path.join(REFERENCE_ROOT, "RoMaV2", "src")
    if src not in sys.path:
        sys.path.insert(0, src)
    import romav2
    romav2.RoMaV2.load_state_dict = lambda self, sd, *a, **kw: None       # the seeded random initialisation IS the model
    # upstream's wrapper module, imported under a synthetic parent so that /root/reference/__init__.py (GUI registration) never runs
    parent = types.ModuleType("_lfd_upstream_m")
    parent.__path__ = [REFERENCE_ROOT]
    sys.modules["_lfd_upstream_m"] = parent
    up = importlib.import_module("_lfd_upstream_m.core.matcher")
    return romav2, up


def _images(seed, sizes):
    from PIL import Image
    rs = np.random.RandomState(seed)
    out = []
    for (w, h) in sizes:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        base = np.stack([np.sin(xx / 17.0 + seed) + np.cos(yy / 23.0), np.cos(xx / 11.0) * np.sin(yy / 13.0 + seed), np.sin((xx + yy) / 29.0)], -1)
        img = (base * 60 + 128 + rs.normal(0, 12, (h, w, 3))).clip(0, 255).astype(np.uint8)
        out.append(Image.fromarray(img))
    return out


def _sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def _build(cls, setting, **kw):
    torch.manual_seed(1234)                 # the same weights in both wrappers
    np.random.seed(0)
    return cls(device="cpu", setting=setting, **kw)


class _Cache:
    """the FeatureCache interface core/matcher.py's keyed descriptor talks to (core/scheduler.py::FeatureCache), kept trivial here"""

    def __init__(self):
        self.d, self.hits, self.misses = {}, 0, 0

    def lookup(self, key, variant):
        v = self.d.get((key, variant))
        if v is not None:
            self.hits += 1
        return v

    def store(self, key, value, variant):
        self.misses += 1
        self.d[(key, variant)] = value
        return value

    def get_or_compute(self, key, fn, variant=None):
        v = self.lookup(key, variant)
        return v if v is not None else self.store(key, fn(), variant)


def check_setting(up, ours_mod, setting, report):
    t0 = time.time()
    ref_m = _build(up.RomaMatcher, setting)
    imA, imB1, imB2, imB3 = _images(3, [(97, 75), (120, 80), (64, 96), (97, 75)])
    with torch.inference_mode():
        expect = ref_m.match_grids_batch(imA, [imB1, imB2, imB3])
    H, W = expect[0][1].shape
    entry = {"grid": [int(H), int(W)], "w_resized": int(ref_m.w_resized), "h_resized": int(ref_m.h_resized), "sample_thresh": float(ref_m.sample_thresh),
             "warp_sha256": [_sha(w) for w, _ in expect], "cert_sha256": [_sha(c) for _, c in expect], "checks": {}}
    # 1. upstream's four-channel layout
    m4 = _build(ours_mod.RomaMatcher, setting, two_channel=False)
    assert (m4.w_resized, m4.h_resized, m4.sample_thresh) == (ref_m.w_resized, ref_m.h_resized, ref_m.sample_thresh)
    got = m4.match_grids_batch(imA, [imB1, imB2, imB3])
    for (w, c), (we, ce) in zip(got, expect):
        assert w.shape == we.shape and torch.equal(w, we) and torch.equal(c, ce), f"{setting}: four-channel output differs from upstream"
    entry["checks"]["four_channel"] = "bit-identical"
    # 2. the two-channel layout the kernels consume + the A-grid as axes
    m2 = _build(ours_mod.RomaMatcher, setting)
    got2 = m2.match_grids_batch(imA, [imB1, imB2, imB3])
    ax, ay = m2.reference_axes(H, W)
    for (w, c), (we, ce) in zip(got2, expect):
        assert torch.equal(w, we[..., 2:4]) and torch.equal(c, ce)
        assert torch.equal(ax.view(1, W).expand(H, W), we[..., 0]) and torch.equal(ay.view(H, 1).expand(H, W), we[..., 1])
    entry["checks"]["two_channel_plus_axes"] = "bit-identical"
    # 3. shared backbone features: a second reference whose neighbour was the first one's reference; keyed descriptor swapped in for model.f
    cache = _Cache()
    m2.set_feature_cache(cache)
    f_before = m2.model.f
    a = m2.match_grids_batch(imA, [imB1, imB2, imB3], keys=(10, [11, 12, 13]))
    b = m2.match_grids_batch(imB1, [imA, imB2], keys=(11, [10, 12]))
    assert m2.model.f is f_before, "model.f was not restored after a keyed call"
    with torch.inference_mode():
        expect_b = ref_m.match_grids_batch(imB1, [imA, imB2])
    for (w, c), (we, ce) in zip(a, expect):
        assert torch.equal(w, we[..., 2:4]) and torch.equal(c, ce), f"{setting}: keyed call differs from upstream"
    for (w, c), (we, ce) in zip(b, expect_b):
        assert torch.equal(w, we[..., 2:4]) and torch.equal(c, ce), f"{setting}: features served from the cache change the result"
    assert cache.hits == 3 and cache.misses == 4, (cache.hits, cache.misses)        # second call: reference + both neighbours from the cache
    entry["checks"]["share_features"] = "bit-identical; 3 of 7 backbone passes served from the cache"
    m2.set_feature_cache(None)
    # 4. several pairs per forward
    m3 = _build(ours_mod.RomaMatcher, setting, pairs_per_forward=3)
    got3 = m3.match_grids_batch(imA, [imB1, imB2, imB3])
    exact = all(torch.equal(w, we[..., 2:4]) and torch.equal(c, ce) for (w, c), (we, ce) in zip(got3, expect))
    dw = max(float((w - we[..., 2:4]).abs().max()) for (w, _), (we, _) in zip(got3, expect))
    dc = max(float((c - ce).abs().max()) for (_, c), (_, ce) in zip(got3, expect))
    entry["checks"]["pairs_per_forward_3"] = "bit-identical" if exact else f"NOT bit-identical: max |d warp| {dw:.3e}, max |d cert| {dc:.3e} (batched bf16 arithmetic rounds differently)"
    assert dw < 5e-2 and dc < 5e-2, f"{setting}: pairs_per_forward=3 is off by more than rounding ({dw}, {dc})"
    for m in (ref_m, m4, m2, m3):
        m.close()
    entry["seconds"] = round(time.time() - t0, 1)
    report[setting] = entry
    print(f"{setting}: grid {H}x{W}: " + "; ".join(f"{k}: {v}" for k, v in entry["checks"].items()) + f"  ({entry['seconds']} s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--settings", nargs="+", default=["turbo", "fast", "high"])
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    romav2, up = install_stubs()
    ours_mod = importlib.import_module("lichtfeld_densification_plugin_amd.core.matcher")
    assert ours_mod._import_romav2() is romav2.RoMaV2, "this package's shim must wrap the very class upstream wraps"
    report = {"versions": {"torch": torch.__version__, "numpy": np.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability()},
              "upstream": {"matcher": "core/matcher.py:74-211", "model": "RoMaV2/src/romav2/romav2.py:93-113,163-269,404-428"},
              "stubs": ["torch.hub.load (DINOv3 backbone interface)", "torchvision.models.vgg19_bn (VGG19-BN features)",
                        "torch.hub.load_state_dict_from_url + RoMaV2.load_state_dict (seeded random weights)", "lichtfeld.log"]}
    for s in args.settings:
        check_setting(up, ours_mod, s, report)
    if args.write:
        with open(os.path.join(HERE, "g8_matcher_contract.json"), "w") as fh:
            json.dump(report, fh, indent=1)
        print("wrote g8_matcher_contract.json")


if __name__ == "__main__":
    main()
