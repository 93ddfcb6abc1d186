import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams); dens.seed_rng(0)
cfg = lfd.DensePipelineConfig(output_path="", roma_setting="fast", nns_per_ref=3)
params = hb.make_params(cfg)
nbrs = synthetic.ring_neighbours(185, 0, 3)
s = synthetic.synth_reference(cams, 0, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000, cert_mode="smooth", device=dev)
r = hb.ReferenceInputs(ref_cam=0, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image)
b = hb.PreparedBatch([r], 512, 512)
best, _ = dens.aggregate(b, params)
w = torch.clamp(best[0], max=0.9); w[:2] = 0; w[-2:] = 0; w[:, :2] = 0; w[:, -2:] = 0
s_up = float(w.cpu().sum())
for name, so in (("device sum", 0.0), ("handed-in normaliser", s_up)):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for it in range(30):
        torch.cuda.synchronize()
        ev[0].record()
        sel = dens.select_samples(best[0], 10000, cap=0.9, border=2, tiles=24, s_override=so)
        ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(name, "median ms per select_samples call (incl. read-back)", float(np.median(ts)), "cells", int(sel.numel()))
