"""How often does the device's normaliser of the coverage sampling differ from upstream's, and does it matter?

Upstream divides the weights by ``weights.sum()`` - a torch CPU f32 reduction whose rounding depends on the vector ISA and on the
number of threads (core/sampling.py:27).  The device uses the correctly rounded exact sum.  Over seeded synthetic certainty maps this
script counts (a) maps whose two normalisers differ (by how many f32 ulps), (b) maps where the difference changes ANY selected cell,
drawing with the same legacy MT19937 stream (oracle.select_samples with s_override).  CPU only; writes a table to stdout.
    python profiles/s_normaliser.py [maps_per_size]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import densify_oracle as orc      # checker
from lichtfeld_densification_plugin_amd import synthetic

n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cams = synthetic.ring_cameras(60, seed=0)


def weights_of(best, cap=0.9, border=2):
    h, w = best.shape
    yy, xx = np.mgrid[0:h, 0:w]
    inside = (xx >= border) & (xx <= w - 1 - border) & (yy >= border) & (yy <= h - 1 - border)
    return (np.minimum(best, np.float32(cap)) * inside.astype(np.float32)).astype(np.float32)


print(f"torch {torch.__version__}, numpy {np.__version__}, {os.cpu_count()} host CPUs")
for (H, W, M) in ((320, 320, 10000), (512, 512, 10000)):
    for mode in ("smooth", "beta", "tiefree"):
        differ = changed = 0
        ulps = []
        sizes_changed = []
        for i in range(n_maps):
            s = synthetic.synth_reference(cams, (7 * i) % 60, synthetic.ring_neighbours(60, (7 * i) % 60, 3), H, W, W, H, noise_px=0.3,
                                          channels=2, seed=500 + i, cert_mode=mode)
            best = np.maximum(s.cert.numpy(), np.float32(0.2)).max(axis=0)          # floor, then per-cell maximum over the neighbours
            wts = weights_of(best)
            s_exact = np.float32(wts.astype(np.float64).sum())                      # what the device computes
            flat = torch.from_numpy(wts.reshape(-1))
            s_variants = set()
            for threads in (1, 2, 4, 8):
                torch.set_num_threads(threads)
                s_variants.add(np.float32(flat.sum().item()))
            torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
            s_torch = np.float32(flat.sum().item())
            if s_torch != s_exact or len(s_variants) > 1:
                differ += 1
                ulps.append(abs(int(np.float32(s_torch).view(np.int32)) - int(np.float32(s_exact).view(np.int32))))
                a = orc.select_samples(best, M, rng=np.random.RandomState(i), s_override=float(s_exact))
                b = orc.select_samples(best, M, rng=np.random.RandomState(i), s_override=float(s_torch))
                if a.shape != b.shape or not np.array_equal(a, b):
                    changed += 1
                    sizes_changed.append(int(np.setxor1d(a, b).size))
        print(f"{H}x{W} M={M} certainty '{mode}': {differ}/{n_maps} maps with s(torch) != s(exact)"
              f"{' (max %d ulp)' % max(ulps) if ulps else ''}; selection changed in {changed} of them"
              f"{' (cells differing: %s)' % sizes_changed[:8] if sizes_changed else ''}", flush=True)
