"""The chained sampled call (lfd_triangulate_sampled_chain: 16 references of the bench workload per call on ONE MT19937 stream) alone, for
`rocprofv3 --kernel-trace --stats -- python3 profiles/chain_profile.py`: which kernel the 0.06 ms per reference go to."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lichtfeld_densification_plugin_amd as lfd                                   # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic                           # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb              # noqa: E402


def main(R=16, reps=20, H=512, W=512, k=3, M=10000):
    dev = torch.device("cuda:0")
    dens = hb.HipDensifier(dev)
    cams = synthetic.ring_cameras(185, seed=0)
    dens.upload_cameras(cams)
    refs = []
    for r in range(R):
        nbrs = synthetic.ring_neighbours(185, r, k)
        s = synthetic.synth_reference(cams, r, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=r, device=dev)
        refs.append(hb.ReferenceInputs(ref_cam=r, nbr_cams=nbrs, cert=[s.cert[j] for j in range(k)], warp=[s.warp[j] for j in range(k)], image=s.image))
    cfg = lfd.DensePipelineConfig(output_path="", matches_per_ref=M, nns_per_ref=k)
    params = hb.make_params(cfg)
    batch = hb.PreparedBatch(refs, W, H)
    cap = M + 24 * 24 + 64
    outs = [hb.OutputBuffers(cap * R, R, k, dev) for _ in range(2)]
    for warm in (True, False):
        dens.seed_rng(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        dens.launch_sampled_chain(batch, params, M, outs[0])
        for i in range(reps):
            if i + 1 < reps:
                dens.launch_sampled_chain(batch, params, M, outs[(i + 1) & 1])
            n += outs[i & 1].collect(indexed=True, check_selection=True).count
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"chained: {dt / (reps * R) * 1e3:.4f} ms per reference, {n / (reps * R):.0f} points per reference")
    dens.close()


if __name__ == "__main__":
    main()
