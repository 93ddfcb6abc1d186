#!/usr/bin/env python3
"""Benchmark of the dense-initialisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused HIP kernel (certainty floor -> arg-max over neighbours -> Sampson
-> DLT -> reprojection / cheirality / parallax -> colour -> ordered compaction) over one batch of
synthetic RoMa outputs that is already resident in HBM: R reference views x k neighbours at the
`fast` preset's 512x512 grid (MipNeRF360 `garden` geometry: 185 cameras on a ring, 1297x840 images),
default filter thresholds.  Metric: triangulated (surviving) points per second, whole job.

For N > 1 the references are dealt round-robin to the ranks (one process per GPU, no data-path
collective inside the timed region: the path shards by reference view) and the survivors are
all-gathered once after the timed region as a correctness check of the multi-GPU path.

Prints ONE JSON line on rank 0 (see the contract in the task description) with two extra objects:
`roofline` (HBM roofline of the fused kernel, measured live with HIP events) and `cpu_baseline` (the
NumPy oracle = a faithful port of the upstream CPU path, timed on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import lichtfeld_densification_plugin_amd as lfd  # noqa: E402
from lichtfeld_densification_plugin_amd import synthetic  # noqa: E402
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--refs", type=int, default=64, help="reference views resident per GPU (per launch)")
    ap.add_argument("--k", type=int, default=3, help="neighbours per reference (GUI default 3)")
    ap.add_argument("--preset", default="fast", choices=sorted(synthetic.ROMA_PRESETS))
    ap.add_argument("--noise-px", type=float, default=0.5)
    ap.add_argument("--outliers", type=float, default=0.05)
    ap.add_argument("--cpu-sample-refs", type=int, default=24,
                    help="references timed on the CPU oracle, ~0.5 s each (0 = skip)")
    ap.add_argument("--spinup-s", type=float, default=0.25, help="untimed spin-up (launch + sync in a loop) before the warm-up steps")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a rocprofv3 --pmc run of this same command (profiles/)")
    return ap.parse_args()


def _committed_pmc(args):
    """profiles/traffic.json (rocprofv3 --pmc passes of this same command) when it was taken on this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        w = t.get("workload", {})
        if (w.get("refs"), w.get("k"), w.get("preset")) == (args.refs, args.k, args.preset):
            return t
    except Exception:
        pass
    return None


def traffic_bytes(args):
    """HBM bytes per launch of the fused kernel: --traffic-bytes, else the committed PMC measurement
    (profiles/traffic.json) when it was taken on this same workload, else null."""
    if args.traffic_bytes is not None:
        return args.traffic_bytes
    t = _committed_pmc(args)
    return float(t["traffic_bytes"]) if t else None


def valu_busy_frac(args):
    """Fraction of the SIMD time the vector ALU was issuing (same PMC run): the kernel's actual bound."""
    t = _committed_pmc(args)
    return (t.get("valu") or {}).get("valu_busy_frac") if t else None


def build_workload(args, rank, world, dev):
    """Global reference list dealt round-robin; this rank generates and keeps only its share."""
    h_lr, w_lr, H, W = synthetic.ROMA_PRESETS[args.preset]
    n_cams = 185
    cams = synthetic.ring_cameras(n_cams, seed=0)
    total_refs = args.refs * world
    ref_ids = [(i * 3) % n_cams for i in range(total_refs)]          # spread over the ring
    mine = [i for i in range(total_refs) if i % world == rank]
    refs, srefs = [], []
    for gi in mine:
        ref = ref_ids[gi]
        nbrs = synthetic.ring_neighbours(n_cams, ref, args.k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, w_lr, h_lr, noise_px=args.noise_px,
                                      outlier_frac=args.outliers, channels=2, seed=1000 + gi, cert_mode="smooth",
                                      device=dev)
        srefs.append(s)
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(args.k)],
                                       warp=[s.warp[j] for j in range(args.k)], image=s.image))
    return cams, refs, srefs, (H, W, w_lr, h_lr), mine


def sampled_mode_rate(args, dens, refs, dims, cfg, n_refs=16):
    """Upstream-equivalent mode, reference after reference as the pipeline runs it: aggregate kernel ->
    on-device coverage sampling (M=10000) -> indexed kernels in one asynchronous call (lfd_triangulate_sampled),
    then the read-back of the counts.  Reported next to the headline (dense) number, not instead of it."""
    H, W, wm, hm = dims
    params = hb.make_params(cfg)
    todo = refs[:n_refs]
    batches = [hb.PreparedBatch([r], wm, hm) for r in todo]
    pts = 0
    for warm in (True, False):
        dens.seed_rng(cfg.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts = 0
        for b in batches:       # what core/pipeline.py runs per reference: one fused call, one read-back
            out = dens.triangulate_sampled(b, params, cfg.matches_per_ref, cap=0.9, border=2, tiles=24)
            pts += out.count
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res = {"refs_per_s": len(todo) / dt, "pairs_per_s": len(todo) * args.k / dt, "points_per_s": pts / dt,
           "ms_per_reference": dt / len(todo) * 1e3, "matches_per_ref": cfg.matches_per_ref, "references_timed": len(todo)}
    # the call is asynchronous: a driver that launches reference i+1 before it reads reference i back (the MT19937 stream
    # is advanced on the device, in stream order) hides the host side and the read-back
    cap = cfg.matches_per_ref + 24 * 24 + 64
    bufs = [hb.OutputBuffers(cap, 1, args.k, dens.device) for _ in range(2)]
    for warm in (True, False):
        dens.seed_rng(cfg.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts2 = 0
        dens.launch_sampled(batches[0], params, cfg.matches_per_ref, bufs[0], cap=0.9, border=2, tiles=24)
        for i in range(len(batches)):
            if i + 1 < len(batches):
                dens.launch_sampled(batches[i + 1], params, cfg.matches_per_ref, bufs[(i + 1) & 1], cap=0.9, border=2, tiles=24)
            pts2 += bufs[i & 1].collect(indexed=True, check_selection=True).count
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
    assert pts2 == pts, (pts2, pts)
    res["pipelined_ms_per_reference"] = dt2 / len(todo) * 1e3
    return res


def _event_ms(fn, reps):
    """Mean HIP-event time of fn() on torch's current stream (the stream the library launches on)."""
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def secondary_kernels(args, dens, batch, refs, dims, cfg, res):
    """The other kernels of the path (SURVEY section 8 rows F1, S, F4 and the N1 writers), each against its own
    algorithmic bytes: aggregate (4k read + 5 written per cell), the PLY / points3D packers (28 read + 15 / 43
    written per point); the selection and indexed kernels are latency-bound single-workgroup kernels, reported in
    ms per reference view."""
    H, W, wm, hm = dims
    params = hb.make_params(cfg)
    out = {}
    best = torch.empty((batch.n_refs, H, W), dtype=torch.float32, device=dens.device)
    slot = torch.empty((batch.n_refs, H, W), dtype=torch.uint8, device=dens.device)
    ms = _event_ms(lambda: dens.launch_aggregate(batch, params, best, slot), 20)
    cells = batch.n_refs * H * W
    out["lfd_aggregate_kernel"] = {"ms": ms, "GB/s": cells * (4 * args.k + 5) / (ms * 1e-3) / 1e9, "bytes_per_cell": 4 * args.k + 5}
    n = int(res.xyz.shape[0])
    ms = _event_ms(lambda: dens.pack_ply(res.xyz, res.rgb), 10)
    out["lfd_pack_ply_kernel"] = {"ms": ms, "GB/s": n * 43 / (ms * 1e-3) / 1e9, "bytes_per_point": 43, "points": n}
    ms = _event_ms(lambda: dens.pack_points3d(res.xyz, res.rgb, res.err), 10)
    out["lfd_pack_points3d_kernel"] = {"ms": ms, "GB/s": n * 71 / (ms * 1e-3) / 1e9, "bytes_per_point": 71, "points": n}
    one = hb.PreparedBatch([refs[0]], wm, hm)
    b1, _ = dens.aggregate(one, params)
    dens.seed_rng(cfg.seed)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        sel = dens.select_samples(b1[0], cfg.matches_per_ref, cap=0.9, border=2, tiles=24)      # synchronises (count to host)
    out["lfd_select_filter_kernel"] = {"ms_per_reference_incl_sync": (time.perf_counter() - t0) / reps * 1e3, "selected": int(sel.numel())}
    ob = hb.OutputBuffers(int(sel.numel()), 1, args.k, dens.device)
    ms = _event_ms(lambda: dens.launch_indexed(one, params, sel, [0, int(sel.numel())], ob), 10)
    out["lfd_indexed_kernel"] = {"ms_per_reference": ms}
    return out


def d2h_inclusive_rate(dens, batch, params, out, dev, reps=3):
    """One launch + copy of the survivors (xyz, rgb, err) into pinned host memory, as a caller that wants NumPy
    arrays pays it.  The C-ABI hands over device pointers, so this is NOT the headline value: it is the
    PCIe-inclusive figure for reference."""
    n = int(out.ref_offsets[-1].item())
    host = [torch.empty((n, 3), dtype=torch.float32).pin_memory(), torch.empty((n, 3), dtype=torch.float32).pin_memory(),
            torch.empty((n,), dtype=torch.float32).pin_memory()]
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        dens.launch_dense(batch, params, out)
        for h, d in zip(host, (out.xyz, out.rgb, out.err)):
            h.copy_(d[:n], non_blocking=True)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"points_per_s": n / best, "ms": best * 1e3, "bytes_to_host": n * 28}


def cpu_baseline(args, cams, srefs, dims, cfg):
    """The oracle (NumPy restatement of upstream's CPU path: same LAPACK batched f32 SVD, same dtype
    ladder) on every cell of a few references of this very workload, single-threaded."""
    if args.cpu_sample_refs <= 0:
        return None
    from oracle import densify_oracle as orc     # checker / baseline only
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                              # pragma: no cover
        threadpool_limits = None
    H, W, wm, hm = dims
    params = orc.OracleParams(certainty_thresh=cfg.certainty_thresh, reproj_thresh=cfg.reproj_thresh,
                              sampson_thresh=cfg.sampson_thresh, min_parallax_deg=cfg.min_parallax_deg)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))

    def oc(c):
        return orc.OracleCamera(K=c.K, R=c.R, t=c.t, P=c.P, C=c.C, width=c.width, height=c.height)

    sample = srefs[:args.cpu_sample_refs]
    host = [([s.cert[j].cpu().numpy() for j in range(args.k)], [s.warp[j].cpu().numpy() for j in range(args.k)],
             s.image.cpu().numpy(), oc(cams[s.ref_index]), [oc(cams[n]) for n in s.nbr_indices]) for s in sample]
    pts = 0
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        for certs, warps, img, ca, cbs in host:
            pts += orc.triangulate_dense(certs, warps, img, ca, cbs, wm, hm, params, axes=axes)["xyz"].shape[0]
    dt = time.perf_counter() - t0
    if ctx is not None:
        ctx.restore_original_limits() if hasattr(ctx, "restore_original_limits") else None
    return {"value": pts / dt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"{len(sample)} reference views x {args.k} neighbours x {H}x{W} cells of this workload "
                      f"({pts} survivors in {dt:.1f} s, NumPy oracle, BLAS threads limited to 1)",
            "pairs_per_s": len(sample) * args.k / dt}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("LFD_BENCH_FORCE_DIST"):     # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cams, refs, srefs, dims, mine = build_workload(args, rank, world, dev)
    H, W, wm, hm = dims
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)   # GUI defaults
    params = hb.make_params(cfg)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, wm, hm)
    out = hb.OutputBuffers(len(refs) * H * W, len(refs), args.k, dev, with_cell=False, with_segments=False)   # upstream emits xyz, rgb, err only

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # untimed spin-up (clocks, first-touch of the output pages), then the W warm-up steps
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:
        dens.launch_dense(batch, params, out)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        dens.launch_dense(batch, params, out)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    host_t = []
    for a, b in ev:
        a.record()                      # same stream the kernel is launched on (torch's current stream)
        dens.launch_dense(batch, params, out)
        b.record()
        host_t.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("LFD_BENCH_DEBUG"):
        print(f"[rank {rank}] enqueue done at {t_enq * 1e3:.2f} ms, first 5 host stamps {[round(x * 1e3, 2) for x in host_t[:5]]}", file=sys.stderr)
    per_launch = [a.elapsed_time(b) for a, b in ev]
    kernel_ms = float(np.mean(per_launch))
    if os.environ.get("LFD_BENCH_DEBUG"):
        print(f"[rank {rank}] per-launch ms: min {min(per_launch):.3f} max {max(per_launch):.3f} mean {kernel_ms:.3f}; "
              f"wall {elapsed * 1e3:.2f} ms for {args.steps} steps", file=sys.stderr)
    dens.check_launches()
    res = out.collect()
    n_pts = res.count

    stats = torch.tensor([elapsed, float(n_pts), kernel_ms], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        total_pts = float(tsum[1].item())
        # the one exchange step of the path: ordered all-gather of the survivors (outside the timed region)
        from lichtfeld_densification_plugin_amd.core import distributed as lfd_dist
        gathered = lfd_dist.all_gather_points(res.xyz, res.rgb, res.err, dist)
        assert gathered[0].shape[0] == int(total_pts), (gathered[0].shape, total_pts)
    else:
        total_pts = float(n_pts)

    if rank == 0:
        cells = len(refs) * H * W
        s_frac = n_pts / cells
        bytes_per_cell = 4 * args.k + 11 + 28 * s_frac
        algo_bytes = cells * bytes_per_cell
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        value = total_pts * args.steps / elapsed
        line = {
            "metric": "triangulated points/sec + pairs/sec, MipNeRF360 garden @fast, 1/2/4/8 GPU",     # BASELINE.json's metric; value = points/s, pairs_per_s beside it
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (+f64 Sampson/DLT solve)", "data": "synthetic",
            "config": {"workload": f"garden-like ring of 185 cameras 1297x840, `{args.preset}` grid {H}x{W}, "
                                   f"{args.refs} reference views x {args.k} neighbours resident per GPU, "
                                   f"default thresholds (certainty 0.2 / sampson 5.0 / reproj 0.8 / parallax 0.5 deg), "
                                   f"noise {args.noise_px} px, {args.outliers:.0%} outliers",
                       "kernel": "fused dense filter+triangulate kernel (lfd_dense_kernel)", "mode": "dense", "refs_per_gpu": args.refs, "neighbours": args.k, "grid": [H, W],
                       "sharding": f"references round-robin over {world} rank(s)"},
            "pairs_per_s": args.refs * world * args.k * args.steps / elapsed,
            "cells_per_s": cells * world * args.steps / elapsed,
            "survivor_fraction": s_frac,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic_bytes(args),
                         "kernel": "lfd_dense_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_cell": bytes_per_cell},
        }
        line["roofline"]["valu_busy_frac"] = valu_busy_frac(args)     # the f64 geometry makes the kernel vector-ALU-bound, not HBM-bound
        line["roofline"]["traffic_over_algorithmic"] = (line["roofline"]["traffic"] / algo_bytes) if line["roofline"]["traffic"] else None
        line["with_d2h"] = d2h_inclusive_rate(dens, batch, params, out, dev)
        line["sampled_mode"] = sampled_mode_rate(args, dens, refs, dims, cfg)
        line["secondary_kernels"] = secondary_kernels(args, dens, batch, refs, dims, cfg, res)
        base = cpu_baseline(args, cams, srefs, dims, cfg)
        if base is not None:
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    dens.close()


if __name__ == "__main__":
    main()
