"""End-to-end pairs/s of run_dense_pipeline with a matcher of FIXED latency (no RoMa weights offline): what the pair scheduler's
feature sharing (core/scheduler.py) is worth.  The stand-in matcher spends T_F ms per backbone pass it actually makes (cached
cameras cost nothing) and T_M ms per (reference, neighbour) pair for the rest of the forward, and returns synthetic maps already
on the GPU; the triangulation runs for real (sampled mode, fused call, launch-ahead).
    python profiles/scheduler_gain.py [n_cams] [T_F ms] [T_M ms]
The latencies are ASSUMPTIONS (defaults 12 ms per DINOv3 ViT-L/16 pass and 20 ms per matcher + refiners pass at 512^2, roughly the
23-33 pairs/s upstream's own test asserts on its GPU, RoMaV2/tests/test_bidirectional.py:47,87); the ratio is what matters."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import pipeline as pl, selection

n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 60
T_F = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
T_M = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
H = W = 256
dev = torch.device("cuda:0")


class FixedLatencyMatcher:
    supports_feature_keys = True
    accepts_device_images = True

    def __init__(self, cams, cache_on):
        self.w_resized = self.h_resized = W
        self.sample_thresh = 0.9
        self.cams, self.cache, self.cache_on = cams, None, cache_on
        self.backbone = 0
        self.maps = {}

    def set_feature_cache(self, cache):
        self.cache = cache if self.cache_on else None

    def _features(self, key):
        def compute():
            self.backbone += 1
            time.sleep(T_F * 1e-3)
            return key
        return self.cache.get_or_compute(key, compute) if self.cache is not None else compute()

    def match_grids_batch(self, imA, imB_list, keys=None):
        ref, nbrs = keys
        self._features(ref)
        out = []
        if ref not in self.maps:
            s = synthetic.synth_reference(self.cams, ref, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=ref, device=dev)
            self.maps[ref] = [(s.warp[j], s.cert[j]) for j in range(len(nbrs))]
        for j, n in enumerate(nbrs):
            self._features(n)
            time.sleep(T_M * 1e-3)
            out.append(self.maps[ref][j])
        return out

    def close(self):
        pass


def main():
    from PIL import Image
    tmp = tempfile.mkdtemp()
    cams = synthetic.ring_cameras(n_cams, seed=0)
    for i, c in enumerate(cams):
        c.image_path = os.path.join(tmp, f"{i:03d}.png")
        Image.fromarray(synthetic.synth_image(H, W, i).numpy()).save(c.image_path)
    flat = np.stack([c.flat_pose() for c in cams])
    print(f"{n_cams} cameras, backbone pass {T_F} ms, rest of a pair's forward {T_M} ms (assumed), grid {H}x{W}")
    for label, frac, k in (("GUI defaults", 0.8, 3), ("CLI defaults", 0.75, 4), ("config 4", 0.3, 8)):
        refs = selection.select_cameras_kcenters(flat, max(1, round(frac * n_cams)))
        nn = selection.nearest_neighbors(flat, k)
        res = {}
        for share in (False, True):
            m = FixedLatencyMatcher(cams, share)
            cfg = lfd.DensePipelineConfig(output_path=os.path.join(tmp, "o.ply"), nns_per_ref=k, viz_interval=0, share_features=share)
            pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=m)                       # warm (maps generated, images decoded)
            m.backbone = 0
            t0 = time.perf_counter()
            r = pl.run_dense_pipeline(cams, refs, nn, cfg, matcher=m)
            dt = time.perf_counter() - t0
            res[share] = (r.pairs_matched / dt, m.backbone, r.xyz.shape[0], dt)
        (p0, b0, n0, t0_), (p1, b1, n1, t1_) = res[False], res[True]
        assert n0 == n1
        print(f"{label:13s} refs {len(refs):3d} x k {k}: upstream schedule {p0:6.1f} pairs/s ({b0} backbone passes, {t0_:.2f} s)   "
              f"shared features {p1:6.1f} pairs/s ({b1} passes, {t1_:.2f} s)   x{p1 / p0:.2f}   [{n1} points either way]")


if __name__ == "__main__":
    main()
