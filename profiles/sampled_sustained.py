#!/usr/bin/env python3
"""The default sampled path (upstream_normaliser: torch's CPU f32 sum per reference) SUSTAINED, not as a burst: bench.py's leg repeated for many
passes, with torch's intra-op threads as they come (one per visible CPU) and fitted to the container's CPU quota.  Why: torch's OpenMP workers spin
between parallel regions; more spinning threads than the cgroup's quota has cores exhaust the quota within a few milliseconds of every 100 ms period
and the whole process is throttled for the rest of it (profiles/r4/sampled_sustained.txt).  Usage: python profiles/sampled_sustained.py [threads|0] [passes]"""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hostenv  # noqa: E402
from lichtfeld_densification_plugin_amd.core.pipeline import _HotPath  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    quota = hostenv.cpu_quota()
    if threads < 0:
        threads = max(1, int(quota)) if quota else 0
    if threads > 0:
        torch.set_num_threads(threads)
    sys.argv = [sys.argv[0], "--refs", "16"]
    args = bench.parse_args()
    dev = torch.device("cuda:0")
    cams, refs, _srefs, dims, _mine, _tot = bench.build_workload(args, 0, 1, dev)
    H, W, wm, hm = dims
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    hot = _HotPath(cams, cfg, 0.9, wm, hm, dev, dens)
    out = []
    for p in range(passes + 1):
        dens.seed_rng(cfg.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend, fly = [], []
        for r in refs:
            pend.append((r, hot.begin_normaliser(r, None)))
            while len(pend) > 1:
                r0, h0 = pend.pop(0)
                fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
            while len(fly) > 1:
                hot.finish_sampled(fly.pop(0))
        while pend:
            r0, h0 = pend.pop(0)
            fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
        while fly:
            hot.finish_sampled(fly.pop(0))
        torch.cuda.synchronize()
        if p:
            out.append((time.perf_counter() - t0) / len(refs) * 1e3)
    out_s = sorted(out)
    print(f"torch threads {torch.get_num_threads()} (cgroup quota {quota} cores, {os.cpu_count()} CPUs visible, OMP_WAIT_POLICY={os.environ.get('OMP_WAIT_POLICY')}): "
          f"{passes} passes of 16 references, ms per reference: first three {[round(v, 3) for v in out[:3]]}, median {out_s[len(out_s) // 2]:.3f}, "
          f"last three {[round(v, 3) for v in out[-3:]]}, max {out_s[-1]:.3f}")
    dens.close()


if __name__ == "__main__":
    main()
