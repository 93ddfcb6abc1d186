#!/usr/bin/env python3
"""A/B on ONE lease, alternating: the ordered dense kernel (look-back retirement) against the unordered one (lfd_triangulate_dense_segments:
one atomic per tile, tile table) and the consumers that restore raster order from the table.  The bench workload, the kernels' own start / stop
events (lfd_kernel_timing); >= 5 alternations, mean +- spread printed per variant.  Usage: python profiles/ab_segments.py [--passes 6] [--steps 200]
[--workload config2|config4|config5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--workload", default="config2")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--workload", a.workload]
    args = bench.parse_args()
    dev = torch.device("cuda:0")
    cams, refs, srefs, dims, mine, total_refs = bench.build_workload(args, 0, 1, dev)
    H, W, wm, hm = dims
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)
    params = hb.make_params(cfg)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)
    n = len(refs)
    out = hb.OutputBuffers(n * H * W, n, args.k, dev, with_cell=False, with_segments=False)
    tpr = (H * W + 1023) // 1024
    table = torch.zeros((n * tpr, 2), dtype=torch.int32, device=dev)
    counts = torch.zeros((n,), dtype=torch.int64, device=dev)

    def run(kind, steps):
        dens.time_dense_kernels(steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if kind == "ordered":
                dens.launch_dense(batch, params, out)
            else:
                dens.launch_dense_segments(batch, params, out, table, counts)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ms = dens.dense_kernel_times_ms()
        dens.time_dense_kernels(0)
        return float(np.mean(ms)), float(np.percentile(ms, 50)), wall

    for kind in ("ordered", "segments"):          # warm both
        run(kind, 64)
    res = {"ordered": [], "segments": []}
    for p in range(a.passes):
        for kind in ("ordered", "segments"):
            m, p50, wall = run(kind, a.steps)
            res[kind].append(m)
            print(f"pass {p} {kind:9s} kernel_ms mean {m:.4f} p50 {p50:.4f}  wall/step {wall:.4f}", flush=True)
    dens.check_launches()
    for kind, v in res.items():
        print(f"{kind:9s} mean {np.mean(v):.4f} ms  +- {np.std(v):.4f}  (min {np.min(v):.4f}, max {np.max(v):.4f}, {len(v)} passes)")
    print(f"segments / ordered = {np.mean(res['segments']) / np.mean(res['ordered']):.4f}")
    # the consumers: what restoring raster order costs (events on the stream)
    seg = hb.SegmentedOutput(out, table, counts, n, H, W, args.k)
    dst = hb.OutputBuffers(n * H * W, n, args.k, dev, with_cell=False, with_segments=False)
    npts = int(counts.sum().item())

    def ev(fn, reps=10):
        return bench._event_ms(fn, reps)
    import ctypes as C
    lib = dens._lib
    ply = torch.empty((n * H * W * 15,), dtype=torch.uint8, device=dev)
    offs = torch.zeros((n + 1,), dtype=torch.int64, device=dev)
    t_order = ev(lambda: lib.lfd_order_segments(dens._ctx, n, H, W, table.data_ptr(), C.byref(out.c), C.byref(dst.c), dst.ref_offsets.data_ptr()))
    t_pack = ev(lambda: lib.lfd_pack_ply_segments(dens._ctx, n, H, W, table.data_ptr(), out.xyz.data_ptr(), out.rgb.data_ptr(), n * H * W, ply.data_ptr(), offs.data_ptr()))
    dens.launch_dense(batch, params, out)
    res_o = out.collect()
    t_pack_plain = ev(lambda: dens.pack_ply(res_o.xyz, res_o.rgb))
    print(f"{npts} survivors: lfd_order_segments {t_order:.4f} ms ({npts * 56 / t_order / 1e6:.0f} GB/s), lfd_pack_ply_segments {t_pack:.4f} ms "
          f"({npts * 43 / t_pack / 1e6:.0f} GB/s) against lfd_pack_ply on ordered records {t_pack_plain:.4f} ms")
    dens.close()


if __name__ == "__main__":
    main()
