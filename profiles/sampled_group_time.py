#!/usr/bin/env python3
"""Sampled (upstream-equivalent) mode on config 4's shape: ms per reference of the fused multi-reference call for the group sizes a
rank of a 1 / 2 / 4 / 8-rank run would use (56 / N references, at most 16 per call).  One JSON line per rank count."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402

dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev)
dens.upload_cameras(cams)
cfg = lfd.DensePipelineConfig(output_path="", nns_per_ref=8)
params = hb.make_params(cfg)
refs = []
for gi in range(56):
    ref = (gi * 3) % 185
    nbrs = synthetic.ring_neighbours(185, ref, 8)
    s = synthetic.synth_reference(cams, ref, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + gi, cert_mode="smooth", device=dev)
    refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(8)], warp=[s.warp[j] for j in range(8)], image=s.image))
cap = cfg.matches_per_ref + 24 * 24 + 64
for n in (1, 2, 4, 8):
    mine = refs[0::n]
    groups = [mine[i:i + 16] for i in range(0, len(mine), 16)]
    batches = [hb.PreparedBatch(g, 512, 512) for g in groups]
    outs = [hb.OutputBuffers(cap * len(g), len(g), 8, dev) for g in groups]
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts = 0
        for rep in range(4):
            for b, o, g in zip(batches, outs, groups):
                dens.launch_sampled_multi(b, params, cfg.matches_per_ref, o, [(7 * i + 1) & 0xFFFFFFFF for i in range(len(g))], cap=0.9, border=2, tiles=24)
                pts += o.collect(indexed=True, check_selection=True).count
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
    print(json.dumps({"ranks": n, "refs_per_rank": len(mine), "mode": "sampled", "step_ms": dt * 1e3, "points": pts / 4}))
