"""cProfile of the DEFAULT sampled path's per-reference host work (bench.py's `default_mode` loop: begin_normaliser / finish_normaliser / launch_sampled /
finish_sampled of core/hotpath.py over 16 references of the bench workload, 8 passes) - where the ~0.25 ms per reference go on the host."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb, hostenv
from lichtfeld_densification_plugin_amd.core.hotpath import HotPath
hostenv.fit_threads_to_quota()
dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams)
cfg = lfd.DensePipelineConfig(output_path="", roma_setting="fast", nns_per_ref=3)
refs = []
for i in range(16):
    ref = (i * 3) % 185
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, 512, 512, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + i, cert_mode="smooth", device=dev)
    refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image))
hot = HotPath(cams, cfg, 0.9, 512, 512, dev, dens)

def one_pass():
    dens.seed_rng(0)
    pend, fly, pts = [], [], 0
    for r in refs:
        pend.append((r, hot.begin_normaliser(r, None)))
        while len(pend) > 1:
            r0, h0 = pend.pop(0)
            fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
        while len(fly) > 1:
            pts += getattr(hot.finish_sampled(fly.pop(0)), "count", 0)
    while pend:
        r0, h0 = pend.pop(0)
        fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
    while fly:
        pts += getattr(hot.finish_sampled(fly.pop(0)), "count", 0)
    torch.cuda.synchronize()
    return pts

one_pass(); one_pass()
t0 = time.perf_counter(); n = 8
for _ in range(n): one_pass()
print("ms per reference: %.4f" % ((time.perf_counter() - t0) / (n * len(refs)) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(n): one_pass()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
