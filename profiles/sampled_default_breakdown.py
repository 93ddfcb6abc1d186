#!/usr/bin/env python3
"""Where the host time of the DEFAULT sampled path goes per reference (upstream_normaliser=True): the pieces of core/pipeline.py::_HotPath's pipelined
normaliser, timed one by one on this box (torch threads as the process finds them).  Usage: python profiles/sampled_default_breakdown.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402
from lichtfeld_densification_plugin_amd.core.pipeline import _HotPath  # noqa: E402
from lichtfeld_densification_plugin_amd.core.sampling import upstream_weight_sum  # noqa: E402


def t(fn, n=50):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    H = W = 512
    cams = synthetic.ring_cameras(185, seed=0)
    nbrs = synthetic.ring_neighbours(185, 0, 3)
    s = synthetic.synth_reference(cams, 0, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1, cert_mode="smooth", device=dev)
    ref = hb.ReferenceInputs(ref_cam=0, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image)
    cfg = lfd.DensePipelineConfig(output_path="")
    dens = hb.HipDensifier(dev)
    hot = _HotPath(cams, cfg, 0.9, W, H, dev, dens)
    dens.seed_rng(0)
    best_host = torch.rand((H, W)).pin_memory()
    print(f"torch threads {torch.get_num_threads()}, cpu_count {os.cpu_count()}")
    print(f"PreparedBatch with upstream's F (3 pairs)      {t(lambda: hb.PreparedBatch([ref], W, H, cameras=cams)):.3f} ms")
    print(f"PreparedBatch without cameras                  {t(lambda: hb.PreparedBatch([ref], W, H)):.3f} ms")
    print(f"upstream_weight_sum on a pinned host map       {t(lambda: upstream_weight_sum(best_host, cap=0.9, border=2)):.3f} ms")
    print(f"  torch.sum alone (262144 f32)                 {t(lambda: best_host.reshape(-1).sum().item()):.3f} ms")
    h = [None]

    def begin():
        h[0] = hot.begin_normaliser(ref, None)

    def full():
        hh = hot.begin_normaliser(ref, None)
        s_up = hot.finish_normaliser(hh)
        hot.finish_sampled(hot.launch_sampled(ref, None, None, s_override=s_up, batch=hh[0]))
    print(f"begin_normaliser (batch + aggregate + D2H issue)  {t(begin):.3f} ms")
    print(f"one reference end to end, NOT pipelined           {t(full, 30):.3f} ms")
    batch = hb.PreparedBatch([ref], W, H, cameras=cams)
    print(f"launch_sampled + finish_sampled (fused call, exact sum) {t(lambda: hot.finish_sampled(hot.launch_sampled(ref, None, None, batch=batch)), 30):.3f} ms")


if __name__ == "__main__":
    main()
