import os, sys, subprocess, numpy as np
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.getcwd())
    import lichtfeld_densification_plugin_amd as lfd
    from lichtfeld_densification_plugin_amd import synthetic
    from lichtfeld_densification_plugin_amd.core import hip_backend as hb
    dev = torch.device("cuda:0")
    outs = {}
    for name, (n, w, h, f, H, W, wm, hm, k, noise) in {"fast": (185, 1297, 840, 960.0, 512, 512, 512, 512, 3, 0.5), "high": (194, 1237, 822, 915.0, 960, 960, 640, 640, 3, 1.0), "far": (185, 1297, 840, 960.0, 512, 512, 512, 512, 4, 2.0)}.items():
        cams = synthetic.ring_cameras(n, width=w, height=h, focal=f, seed=0)
        refs = []
        for r in (3, 77):
            nb = synthetic.ring_neighbours(n, r, k)
            s = synthetic.synth_reference(cams, r, nb, H, W, wm, hm, noise_px=noise, outlier_frac=0.05, channels=2, seed=r, device=dev, far_depth=400.0 if name == "far" else 25.0)
            refs.append(hb.ReferenceInputs(ref_cam=r, nbr_cams=nb, cert=[s.cert[j] for j in range(k)], warp=[s.warp[j] for j in range(k)], image=s.image))
        dens = hb.HipDensifier(dev); dens.upload_cameras(cams)
        res = dens.triangulate_dense(hb.PreparedBatch(refs, wm, hm, cameras=cams), hb.make_params(lfd.DensePipelineConfig(output_path="")))
        outs[name + "_xyz"] = res.xyz.cpu().numpy(); outs[name + "_cell"] = res.cell.cpu().numpy(); outs[name + "_err"] = res.err.cpu().numpy()
        outs[name + "_off"] = np.asarray(res.ref_offsets)
        dens.close()
    np.savez(sys.argv[1], **outs)
else:
    for v in ("n1", "n2"):
        subprocess.check_call([sys.executable, __file__, f"/tmp/out_{v}.npz"], env=dict(os.environ, LFD_DENSIFY_LIB=f"build/variants/{v}.so"))
    a, b = np.load("/tmp/out_n1.npz"), np.load("/tmp/out_n2.npz")
    for name in ("fast", "high", "far"):
        ca, cb = a[name + "_cell"], b[name + "_cell"]
        same = ca.shape == cb.shape and np.array_equal(ca, cb) and np.array_equal(a[name + "_off"], b[name + "_off"])
        print(name, "survivors", ca.shape[0], cb.shape[0], "same cells:", same)
        if same:
            xa, xb = a[name + "_xyz"], b[name + "_xyz"]
            d = np.abs(xa.astype(np.float64) - xb) / np.maximum(np.abs(xb), 1e-6)
            print("   xyz bit-identical: %.5f %% of the values, max relative difference %.3g; err max abs diff %.3g" % (100.0 * np.mean(xa == xb), d.max(), np.abs(a[name + "_err"] - b[name + "_err"]).max()))
