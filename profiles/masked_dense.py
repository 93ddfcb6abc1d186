"""Dense kernel with reference / neighbour masks (the certainty prologue's general path) vs without."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams)
cfg = lfd.DensePipelineConfig(output_path="", roma_setting="fast", nns_per_ref=3)
params = hb.make_params(cfg)
R, H, W = 32, 512, 512
g = torch.Generator(device="cpu"); g.manual_seed(0)
def blob_mask():
    m = torch.ones((H, W), dtype=torch.uint8)
    for _ in range(6):
        y, x = int(torch.randint(0, H - 80, (1,), generator=g)), int(torch.randint(0, W - 80, (1,), generator=g))
        m[y:y + 60, x:x + 70] = 0
    return m.to(dev)
modes = sys.argv[1:] or ["none", "mask_a", "mask_a+mask_b"]
for mode in modes:
    refs = []
    for i in range(R):
        ref = (3 * i) % 185
        nbrs = synthetic.ring_neighbours(185, ref, 3)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + i, cert_mode="smooth", device=dev)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(3)], warp=[s.warp[j] for j in range(3)], image=s.image,
                                       mask_a=blob_mask() if mode != "none" else None,
                                       mask_b=[blob_mask() for _ in range(3)] if mode == "mask_a+mask_b" else None))
    batch = hb.PreparedBatch(refs, 512, 512)
    out = hb.OutputBuffers(R * H * W, R, 3, dev, with_cell=False, with_segments=False)
    for _ in range(5): dens.launch_dense(batch, params, out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); dens.launch_dense(batch, params, out); b.record()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in ev]
    ms = float(np.mean(per))
    print("   per-launch ms: min %.3f median %.3f max %.3f" % (min(per), float(np.median(per)), max(per)))
    n = int(out.ref_offsets[-1].item())
    print("%-14s %.4f ms per launch (%d refs x 3 x 512^2), %.3e cells/s, survivors %.4f" % (mode, ms, R, R * H * W / (ms * 1e-3), n / (R * H * W)))
