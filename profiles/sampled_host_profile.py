"""Where the per-reference time of the sampled (upstream-equivalent) mode goes on the host side."""
import os, sys, time, torch
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
for ref in range(0, 48, 3):
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + ref, cert_mode="smooth", device=dev)
    refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image))
batches = [hb.PreparedBatch([r], 512, 512) for r in refs]
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for rep in range(3):
    acc.clear()
    for b in batches:
        torch.cuda.synchronize(); t = time.perf_counter()
        best, _ = dens.aggregate(b, params); t = tick("aggregate", t)
        sel = dens.select_samples(best[0], 10000, cap=0.9, border=2, tiles=24); t = tick("select", t)
        ob = hb.OutputBuffers(int(sel.numel()), 1, 3, dev); t = tick("alloc outputs", t)
        dens.launch_indexed(b, params, sel, [0, int(sel.numel())], ob); t = tick("indexed launch+kernel", t)
        dens.check_launches(); t = tick("check_launches", t)
        out = ob.collect(indexed=True); t = tick("collect", t)
print({k: round(v / len(batches) * 1e3, 4) for k, v in acc.items()}, "ms per reference; total", round(sum(acc.values()) / len(batches) * 1e3, 3))
dens.seed_rng(0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in batches:
        out = dens.triangulate_sampled(b, params, 10000, cap=0.9, border=2, tiles=24)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("fused lfd_triangulate_sampled + read-back: %.3f ms per reference" % (dt / len(batches) * 1e3))
