import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams); dens.seed_rng(0)
cfg = lfd.DensePipelineConfig(output_path="", roma_setting="fast", nns_per_ref=3)
params = hb.make_params(cfg)
for ref in (0, 3, 6):
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + ref, cert_mode="smooth", device=dev)
    r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image)
    b = hb.PreparedBatch([r], 512, 512)
    best, _ = dens.aggregate(b, params)
    sel = dens.select_samples(best[0], 10000, cap=0.9, border=2, tiles=24)
    print("selected", int(sel.numel()))
