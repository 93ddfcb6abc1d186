"""Where the host's time goes in the chained default path (upstream's normaliser): per group of 16 references at 512^2 - the batched aggregate + copy,
torch's sums, the fused call, the read-back.  python profiles/chain_host_profile.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lichtfeld_densification_plugin_amd as lfd                                   # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic                           # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb, hostenv     # noqa: E402
from lichtfeld_densification_plugin_amd.core.hotpath import HotPath                # noqa: E402


def main(R=16, groups=12, H=512, W=512, k=3, M=10000, depth_ready=1, depth_fly=1, prebuilt=False):
    hostenv.fit_threads_to_quota()
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
    hot = HotPath(cams, cfg, 0.9, W, H, dev, dens)
    for warm in (True, False):
        dens.seed_rng(0)
        torch.cuda.synchronize()
        t = {"prepare": 0.0, "begin": 0.0, "sums": 0.0, "launch": 0.0, "collect": 0.0}
        t0 = time.perf_counter()
        ready, fly = [], []

        def clock(name, fn):
            a = time.perf_counter()
            r = fn()
            t[name] += time.perf_counter() - a
            return r
        b_pre = hot.prepare_chain(refs, None)
        for _ in range(groups):
            b = b_pre if prebuilt else clock("prepare", lambda: hot.prepare_chain(refs, None))
            ready.append((b, clock("begin", lambda: hot.begin_chain_normalisers(b))))
            while len(ready) > depth_ready:
                b0, s0 = ready.pop(0)
                sums = clock("sums", lambda: hot.finish_chain_normalisers(s0))
                fly.append(clock("launch", lambda: hot.launch_sampled_chain(b0, sums)))
            while len(fly) > depth_fly:
                clock("collect", lambda: hot.finish_sampled(fly.pop(0), check_selection=False))
        while ready:
            b0, s0 = ready.pop(0)
            sums = hot.finish_chain_normalisers(s0)
            fly.append(hot.launch_sampled_chain(b0, sums))
        while fly:
            hot.finish_sampled(fly.pop(0), check_selection=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"ready {depth_ready} fly {depth_fly} prebuilt {prebuilt}: {dt / (groups * R) * 1e3:.4f} ms per reference; host ms per group: " + ", ".join(f"{k_} {v / groups * 1e3:.3f}" for k_, v in t.items()))
    # the sums alone
    x = torch.rand(R, H, W).pin_memory()
    a = time.perf_counter()
    for _ in range(20):
        [float(x[r].reshape(-1).sum()) for r in range(R)]
    print(f"torch sums of {R} maps: {(time.perf_counter() - a) / 20 * 1e3:.3f} ms ({torch.get_num_threads()} threads)")
    dens.close()


if __name__ == "__main__":
    for dr, df in ((1, 1), (2, 1), (1, 2), (2, 2), (3, 3)):
        main(groups=24, depth_ready=dr, depth_fly=df, prebuilt=True)
