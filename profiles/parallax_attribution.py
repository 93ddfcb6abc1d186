#!/usr/bin/env python3
"""VERDICT r3 "weak" 1(a): the kernel tests the parallax as  a.b <= cos(theta_min) (|a| + 1e-12)(|b| + 1e-12)  (csrc/lfd_geometry.hpp) where
upstream normalises both rays and compares the angle (core/geometry.py:113-119: per-component divide by norm + 1e-12, 3-term sum, clip,
arccos, degrees, >=).  How many of the cells the implementation decides differently from the oracle come from that FORMULA, and how many
from X (the f64 null vector against LAPACK's f32 SVD)?

For every flipped cell of the references below, X as the implementation computes it (the per-cell routine itself through
lfd_host_eval_correspondence - the host build of the kernels' source - with `--no-parallax` thresholds so that X is returned whatever the
test says) is put through upstream's exact operation sequence (the oracle's parallax_angle_deg: NumPy f32, the same calls upstream makes).
  formula-caused flip: upstream's sequence ON THE IMPLEMENTATION'S X decides differently from the implementation's own (cross-multiplied) test
  X-caused flip      : both forms agree on the implementation's X; the decision differs from upstream's because X differs
Runs on the GPU kernel when a GPU is present, else on the CPU twin (same per-cell source).  Usage: python profiles/parallax_attribution.py [n_refs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402
from helpers import flip_report, orc  # noqa: E402  (checker)


def attribute(n_refs=4, verbose=True):
    H = W = wm = hm = 512
    cams = synthetic.ring_cameras(185, seed=0)
    on_gpu = torch.cuda.is_available()
    dev = torch.device("cuda:0") if on_gpu else torch.device("cpu")
    dens = hb.HipDensifier(dev) if on_gpu else hb.HostDensifier(0)
    dens.upload_cameras(cams)
    cfg = lfd.DensePipelineConfig(output_path="")
    params = hb.make_params(cfg)
    cfg_np = lfd.DensePipelineConfig(output_path="", min_parallax_deg=0.0, reproj_thresh=1e9, sampson_thresh=0.0)      # X whatever the tests say
    params_np = hb.make_params(cfg_np)
    oparams = orc.OracleParams()
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    tot = dict(cells=0, flips=0, parallax=0, formula=0, x_caused=0, oob=0)
    if verbose:
        print(f"implementation: {'HIP dense kernel' if on_gpu else 'CPU twin (host build of the per-cell source)'}; scene: the bench workload (ring of 185, fast 512^2, k = 3, GUI thresholds)")
    for gi in range(n_refs):
        ref = (gi * 3) % 185
        nbrs = synthetic.ring_neighbours(185, ref, 3)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, wm, hm, noise_px=0.5, outlier_frac=0.05, channels=2, seed=1000 + gi, cert_mode="smooth")
        r = hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j].to(dev) for j in range(3)], warp=[s.warp[j].contiguous().to(dev) for j in range(3)],
                               image=s.image.to(dev))
        out = dens.triangulate_dense(hb.PreparedBatch([r], wm, hm, cameras=cams), params)
        kept = out.cell.cpu().numpy().astype(np.int64)
        rep = flip_report(kept, s, cams, wm, hm, oparams, axes)
        keep_impl = np.zeros(H * W, bool)
        keep_impl[kept] = True
        keep_orc = np.zeros(H * W, bool)
        keep_orc[rep["oracle"].cell] = True
        flipped = np.nonzero(keep_impl != keep_orc)[0]
        reason, _ = orc.classify_flips(flipped, rep["best_k"], rep["agg"], rep["cam_a"], rep["cams_b"], wm, hm, oparams, axes=axes)
        par = flipped[reason == orc.FLIP_REASONS.index("parallax")]
        n_formula = 0
        for cell in par:
            y, x = divmod(int(cell), W)
            j = int(np.asarray(rep["best_k"]).reshape(-1)[cell])
            xbn, ybn = float(s.warp[j][y, x, 0]), float(s.warp[j][y, x, 1])
            o = hb.host_eval_correspondence(cams[ref], cams[nbrs[j]], float(axes[0][x]), float(axes[1][y]), xbn, ybn, wm, hm, params_np)
            X = np.array([[o[0], o[1], o[2]]], np.float32)
            with np.errstate(all="ignore"):
                ang = orc.parallax_angle_deg(np.asarray(cams[ref].C, np.float32), np.asarray(cams[nbrs[j]].C, np.float32), X.copy())
            upstream_form_keeps = bool(ang[0] >= np.float32(oparams.min_parallax_deg))
            # the implementation's own (cross-multiplied) verdict on its X = whether it kept the cell (every other test passed: the flip was
            # classified `parallax`, and the reprojection / depth tests are the same comparisons in both)
            if upstream_form_keeps != bool(keep_impl[cell]):
                n_formula += 1
        tot["cells"] += H * W; tot["flips"] += int(flipped.size); tot["parallax"] += int(par.size); tot["formula"] += n_formula
        tot["x_caused"] += int(par.size) - n_formula; tot["oob"] += int((reason < 0).sum())
        if verbose:
            print(f"ref {ref:3d}: {flipped.size} flips ({(reason < 0).sum()} out of band), {par.size} by the parallax test: {n_formula} formula-caused, {par.size - n_formula} X-caused")
    if verbose:
        print(f"total: {tot['flips']} flips in {tot['cells']} cells ({tot['flips'] / tot['cells']:.2e}), {tot['oob']} out of band; parallax {tot['parallax']}: "
              f"{tot['formula']} formula-caused, {tot['x_caused']} X-caused")
    dens.close()
    return tot


if __name__ == "__main__":
    attribute(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
