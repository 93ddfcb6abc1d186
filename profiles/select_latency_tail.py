"""Per-call latency distribution of the fused sampled call (tail check of the grid-barrier kernel)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams); dens.seed_rng(0)
cfg = lfd.DensePipelineConfig(output_path="", roma_setting="fast", nns_per_ref=3)
params = hb.make_params(cfg)
refs = []
for ref in range(0, 24, 3):
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + ref, cert_mode="smooth", device=dev)
    refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image))
batches = [hb.PreparedBatch([r], 512, 512) for r in refs]
for g in sys.argv[1:] or ["4", "16"]:
    os.environ["LFD_SELECT_WORKGROUPS"] = g
    dens.reload_env()
    lat = []
    for it in range(400):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = dens.triangulate_sampled(batches[it % len(batches)], params, 10000, cap=0.9, border=2, tiles=24)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[20:])
    print("workgroups %-3s: median %.3f  p99 %.3f  max %.3f ms over %d calls" % (g, np.median(lat), np.percentile(lat, 99), lat.max(), lat.size))
