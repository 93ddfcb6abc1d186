"""How many cells of a full-size reference (512^2, k=3, default thresholds) are decided differently by the HIP dense kernel and
by the oracle (= upstream's arithmetic): the cells inside the rounding-noise band of a threshold."""
import os, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from oracle import densify_oracle as orc       # checker
dev = torch.device("cuda:0")
H = W = 512
cams = synthetic.ring_cameras(185, seed=0)
dens = hb.HipDensifier(dev); dens.upload_cameras(cams)
params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
tot_diff = tot_cells = tot_surv = 0
for ref in (0, 30, 60, 90):
    nbrs = synthetic.ring_neighbours(185, ref, 3)
    s = synthetic.synth_reference(cams, ref, nbrs, H, W, 512, 512, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + ref, cert_mode="smooth")
    r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[c.to(dev) for c in s.cert], warp=[w.contiguous().to(dev) for w in s.warp], image=s.image.to(dev))
    out = dens.triangulate_dense(hb.PreparedBatch([r], 512, 512), params)
    oc = lambda c: orc.OracleCamera(K=c.K, R=c.R, t=c.t, P=c.P, C=c.C, width=c.width, height=c.height)
    with np.errstate(all="ignore"):
        d = orc.triangulate_dense([c.numpy() for c in s.cert], [w.numpy() for w in s.warp], s.image.numpy(), oc(cams[ref]), [oc(cams[n]) for n in nbrs],
                                  512, 512, orc.OracleParams(), axes=(orc.identity_axis_scalar(W), orc.identity_axis_scalar(H)))
    cell = out.cell.cpu().numpy()
    only_hip = np.setdiff1d(cell, d["cell"]).size
    only_orc = np.setdiff1d(d["cell"], cell).size
    common = np.intersect1d(cell, d["cell"])
    ph, po = np.searchsorted(cell, common), np.searchsorted(d["cell"], common)
    dx = np.abs(out.xyz.cpu().numpy()[ph] - d["xyz"][po]).max(axis=1) / np.maximum(np.abs(d["xyz"][po]).max(axis=1), 1e-6)
    rgb_equal = np.array_equal(out.rgb.cpu().numpy()[ph], d["rgb"][po])
    print("ref %3d: survivors hip %d oracle %d | kept only by hip %d, only by oracle %d (%.2e of cells) | max rel xyz diff %.2e, rgb bit-identical %s"
          % (ref, cell.size, d["cell"].size, only_hip, only_orc, (only_hip + only_orc) / (H * W), dx.max(), rgb_equal))
    tot_diff += only_hip + only_orc; tot_cells += H * W; tot_surv += cell.size
print("total: %d of %d cells differ (%.2e)" % (tot_diff, tot_cells, tot_diff / tot_cells))
