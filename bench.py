#!/usr/bin/env python3
"""Benchmark of the dense-initialisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused HIP kernel (certainty floor -> arg-max over neighbours -> Sampson
-> DLT -> reprojection / cheirality / parallax -> colour -> ordered compaction) over one batch of
synthetic RoMa outputs that is already resident in HBM: R reference views x k neighbours at the
`fast` preset's 512x512 grid (MipNeRF360 `garden` geometry: 185 cameras on a ring, 1297x840 images),
default filter thresholds.  Metric: triangulated (surviving) points per second, whole job.

For N > 1 (one process per GPU over RCCL) the default job is ONE scene - BASELINE config[3]: 56 reference views x 8 neighbours, dealt
round-robin over the ranks (`--scaling strong --workload config4`) - and the exchange of the survivors is INSIDE the timed region: a step
is the rank's share of the scene in `--exchange-rounds` launches, each launch's survivors packed on the device (15-byte PLY vertex
records by default, `--exchange-records f32` for the 28-byte rows) and handed to the round's asynchronous collective (`--exchange
all_gather` | `gather_to_root`) while the next launch computes, then the wait for the last round.  `value` = the scene's survivors / that
time; `value_compute_only` the same steps without any exchange.  `python bench.py --gpus N` without a launcher starts the N ranks itself
(fresh child processes, before this process touches the GPU); under `torch.distributed.run` it is one of the ranks.  The N = 1 line is
the single-GPU benchmark below, unchanged.

Prints ONE JSON line on rank 0 (see the contract in the task description) with extra objects:
`roofline` (HBM roofline of the fused kernel, measured live with HIP events), `cpu_baseline` (this build's
own C++ restatement - the CPU twin of the C-ABI, host build of the kernels' source - on 1 thread and on all
host threads, the NumPy oracle and upstream's probed Python time quoted beside it) and `parity` (cells of
the timed workload the kernel decides differently from the oracle, each checked against the derived
rounding band).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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

# the synthetic stand-ins of BASELINE.json's configurations (garden: 185 cameras on a ring, 1297x840 images)
WORKLOADS = {
    "config2": dict(refs=64, k=3, preset="fast", n_cams=185, arc=None,
                    what="config[1]: garden @fast, GUI default thresholds, 64 reference views x 3 neighbours resident"),
    "config3": dict(refs=32, k=3, preset="high", n_cams=194, arc=None, width=1237, height=822,
                    what="config[2]: bicycle (194 cameras 1237x822) @high (960x960 grid over 640-px match images), full filter stack, 32 reference views x 3 neighbours resident"),
    "config4": dict(refs=56, k=8, preset="fast", n_cams=185, arc=None,
                    what="config[3]: garden, all cameras, ref-fraction 0.3 -> 56 reference views x 8 neighbours = 448 pairs, sharded"),
    "config5": dict(refs=12, k=8, preset="precise", n_cams=12, arc=0.6,
                    what="config[4]: `precise` 1280x1280 grid, ROI of 12 selected cameras (neighbours clamped to 11 -> 8), sharded"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="BASELINE.json configuration the synthetic workload follows (sets the defaults of --refs / --k / --preset); default: config2 "
                         "(BASELINE configs[1]) on one GPU, config4 (the sharded scene: 56 references x 8 neighbours) on several")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default for --gpus > 1: BASELINE's metric is ONE scene at 1/2/4/8 GPUs): --refs in TOTAL, dealt round-robin over the "
                         "ranks; weak: --refs reference views PER GPU (the work grows with N)")
    ap.add_argument("--exchange", default="all_gather", choices=["all_gather", "gather_to_root"],
                    help="--gpus > 1: the collective the survivors travel in, INSIDE the timed region, in rounds beside the compute")
    ap.add_argument("--exchange-records", default="ply", choices=["ply", "f32"],
                    help="what travels: the 15-byte PLY vertex records packed on the device (the point cloud as it is written: xyz f32 + rgb u8) or the "
                         "28-byte f32 rows (xyz, rgb, err)")
    ap.add_argument("--replicate", default="auto",
                    help="--gpus > 1: references computed by EVERY rank instead of exchanged (recompute instead of communicate): a count, or `auto` - "
                         "core/distributed.py::plan_replication from this run's own measurements (per-reference compute, all-gather bandwidth)")
    ap.add_argument("--exchange-rounds", type=int, default=2, help="rounds a rank's share of the scene is cut into (one launch + one exchange round each)")
    ap.add_argument("--refs", type=int, default=None, help="reference views resident per GPU (weak) / in total (strong)")
    ap.add_argument("--k", type=int, default=None, help="neighbours per reference (GUI default 3)")
    ap.add_argument("--preset", default=None, choices=sorted(synthetic.ROMA_PRESETS))
    ap.add_argument("--cached-batch", action="store_true",
                    help="re-launch ONE prepared batch (descriptor upload and per-pair constants skipped after the first launch) instead of "
                         "alternating two distinct batches - profiling passes that want the dense kernel alone")
    ap.add_argument("--noise-px", type=float, default=0.5)
    ap.add_argument("--outliers", type=float, default=0.05)
    ap.add_argument("--cpu-sample-refs", type=int, default=24,
                    help="references of the workload timed on the CPU twin (0 = skip the cpu_baseline leg)")
    ap.add_argument("--spinup-s", type=float, default=0.25, help="untimed spin-up (bursts of 32 back-to-back launches) before the warm-up steps")
    ap.add_argument("--light", action="store_true", help="headline only: skip the secondary legs (profiling passes)")
    ap.add_argument("--parity-refs", type=int, default=2, help="references of the workload checked cell by cell against the oracle (0 = skip)")
    ap.add_argument("--scene-root", default=os.environ.get("LFD_SCENE_ROOT"),
                    help="a COLMAP scene (images*/ + sparse/0/) for the END-TO-END leg: dense_init through the real RoMa-v2 matcher; needs the `romav2` package "
                         "and its weights (torch hub cache, or LFD_ROMA_WEIGHTS=path/to/romav2.pt); without them `end_to_end` stays null with the reason")
    ap.add_argument("--pipeline-cams", type=int, default=185,
                    help="cameras of the generated on-disk scene of the `pipeline` leg (densify.dense_init end to end, bench_pipeline.py); 0 = skip the leg")
    ap.add_argument("--pipeline-latency-ms", type=float, default=20.0, help="stand-in matcher latency per pair of the leg's second pass (0 = skip that pass)")
    ap.add_argument("--pipeline-size", default="1297x840", help="image size of the generated scene (garden's images_4)")
    ap.add_argument("--dry-run", action="store_true",
                    help="--gpus N: print what every rank WILL do - shard plan, launches and rounds per step, buffer sizes, the collectives in issue "
                         "order - as one JSON object, without touching a GPU or creating a communicator, and exit")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a rocprofv3 --pmc run of this same command (profiles/)")
    args = ap.parse_args()
    many = args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    if args.workload is None:
        args.workload = "config4" if many else "config2"
    if args.scaling is None:
        args.scaling = "strong" if many else "weak"
    w = WORKLOADS[args.workload]
    args.refs = w["refs"] if args.refs is None else args.refs
    args.k = w["k"] if args.k is None else args.k
    args.preset = w["preset"] if args.preset is None else args.preset
    return args


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
    """(HBM bytes per launch of the fused kernel, where the number comes from): --traffic-bytes, else the committed PMC measurement
    (profiles/traffic.json) when it was taken on this same workload, else (null, null).  PMC counters cannot be read from inside
    this process: the figure is never measured by the run that prints it, and the line says so (`traffic_source`)."""
    if args.traffic_bytes is not None:
        return args.traffic_bytes, "--traffic-bytes (rocprofv3 --pmc run of this command by the caller)"
    t = _committed_pmc(args)
    if t:
        return float(t["traffic_bytes"]), "profiles/traffic.json: " + str(t.get("source", "committed rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 + WRITE_SIZE)"))
    return None, None


def valu_busy_frac(args):
    """Fraction of the SIMD time the vector ALU was issuing (same PMC run): the kernel's actual bound."""
    t = _committed_pmc(args)
    return (t.get("valu") or {}).get("valu_busy_frac") if t else None


def valu_roofline(args, kernel_ms):
    """The OTHER bound of the fused kernel, stated beside the HBM one: the issue time of its vector instructions.  Instructions per cell by class
    from the committed rocprofv3 --pmc passes of this command (profiles/traffic.json), real cycles per wave-instruction and the clock the chip
    sustains under this arithmetic from profiles/microbench/clock_under_load.hip; `frac` = issue time / measured kernel time (1.0 = nothing but
    vector issue)."""
    t = _committed_pmc(args)
    v = (t.get("valu") or {}) if t else {}
    if not v.get("issue_ms"):
        return None
    return {"bound": "valu", "instructions_per_cell": v.get("instructions_per_cell"), "f64_per_cell": v.get("f64_per_cell"), "f32_per_cell": v.get("f32_per_cell"),
            "conversions_per_cell": v.get("conversions_per_cell"), "integer_per_cell": v.get("integer_per_cell"),
            "cycles_per_wave_instruction": v.get("cycles_per_wave_instruction"), "sustained_clock_GHz": v.get("sustained_clock_GHz"),
            "issue_cycles_per_cell": v.get("issue_cycles_per_cell"), "issue_ms": v.get("issue_ms"), "frac": v["issue_ms"] / kernel_ms if kernel_ms else None,
            "valu_busy_frac": v.get("valu_busy_frac"), "source": v.get("source")}


def _hwmon_of(dev):
    """The hwmon directory of THIS GPU (sysfs: package power in microwatts, shader clock in Hz), found through the device's PCI address - the box shows the
    cards of other tenants too.  None where sysfs does not say (another driver layout, no permission)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev)
        want = f"{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}"
    except Exception:
        return None
    for card in glob.glob("/sys/class/drm/card*/device"):
        try:
            if os.path.basename(os.path.realpath(card)).lower().startswith(want):
                mons = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                return mons[0] if mons else None
        except OSError:
            continue
    return None


def power_under_kernel(dens, batches, params, out, dev, seconds=1.5):
    """What the third bound of the fused kernel looks like on THIS box while THIS process runs it: the package's power against its cap and the shader clock,
    read from sysfs every ~10 ms beside `seconds` of back-to-back launches (the timed loop's step, outside the timed region).  profiles/r6/clock_power.txt is
    the same measurement with rocm-smi."""
    import threading
    mon = _hwmon_of(dev)
    if mon is None:
        return {"note": "hwmon of this GPU not found in sysfs"}

    def read(name):
        with open(os.path.join(mon, name)) as fh:
            return float(fh.read().strip())
    try:
        cap_w = read("power1_cap") * 1e-6
        read("power1_input"), read("freq1_input")
    except Exception as exc:
        return {"note": f"hwmon not readable ({exc})"}
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append((time.perf_counter(), read("power1_input") * 1e-6, read("freq1_input") * 1e-6))
            except Exception:
                pass
            time.sleep(0.01)
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(64):
            b = batches[n % len(batches)]
            dens.prepare(b, params)
            dens.launch_dense(b, params, out)
            n += 1
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    use = [(w, f) for (t, w, f) in samples if t0 + 0.5 * seconds <= t <= t1]      # the second half: the firmware's averaging window has filled
    if not use:
        return {"note": "no samples"}
    w = float(np.mean([u[0] for u in use]))
    return {"package_W": w, "cap_W": cap_w, "frac_of_cap": w / cap_w if cap_w else None, "sclk_MHz": float(np.mean([u[1] for u in use])), "samples": len(use),
            "launches": n, "ms_per_launch": (t1 - t0) / n * 1e3,
            "source": "sysfs hwmon of this GPU (power1_input, freq1_input, power1_cap), sampled every ~10 ms beside back-to-back launches of the timed loop's step; "
                      "second half of the window",
            "note": "the fused kernel draws the package's power cap or close to it and runs at the shader clock the firmware grants under it (2.4 GHz nominal): its "
                    "speed is energy per cell (DESIGN 4.2; rocm-smi beside 30 000 launches: profiles/r6/clock_power.txt)"}


def build_workload(args, rank, world, dev, positions=None):
    """Global reference list dealt round-robin; this rank generates and keeps only its share (``positions``: these positions of the global
    list instead).  weak scaling: --refs references per rank; strong scaling: --refs in total (BASELINE's metric is ONE scene at 1/2/4/8 GPUs)."""
    h_lr, w_lr, H, W = synthetic.ROMA_PRESETS[args.preset]
    wl = WORKLOADS[args.workload]
    n_cams = wl["n_cams"]
    cams = synthetic.ring_cameras(n_cams, seed=0, **({"arc": wl["arc"]} if wl["arc"] else {}),
                                  **({"width": wl["width"], "height": wl["height"]} if "width" in wl else {}))
    total_refs = args.refs * world if args.scaling == "weak" else args.refs
    step = 3 if n_cams >= 3 * total_refs or n_cams % 3 else 1
    ref_ids = [(i * step) % n_cams for i in range(total_refs)]          # spread over the ring
    mine = [i for i in range(total_refs) if i % world == rank] if positions is None else [int(g) for g in positions]
    k = min(args.k, n_cams - 1)
    refs, srefs = [], []
    for gi in mine:
        ref = ref_ids[gi]
        nbrs = synthetic.ring_neighbours(n_cams, ref, k) if not wl["arc"] else sorted(range(n_cams), key=lambda c: (abs(c - ref), c))[1:k + 1]
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, w_lr, h_lr, noise_px=args.noise_px,
                                      outlier_frac=args.outliers, channels=2, seed=1000 + gi, cert_mode="smooth",
                                      device=dev)
        srefs.append(s)
        pad = int(os.environ.get("LFD_BENCH_PLANE_PAD", "0"))       # experiment: every plane in its own allocation, `pad` bytes apart
        if pad > 0:
            certs, warps = [], []
            for j in range(k):
                _pads.append(torch.empty(pad, dtype=torch.uint8, device=dev)); certs.append(s.cert[j].clone())
                _pads.append(torch.empty(pad, dtype=torch.uint8, device=dev)); warps.append(s.warp[j].clone())
        else:
            certs, warps = [s.cert[j] for j in range(k)], [s.warp[j] for j in range(k)]
        refs.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=certs, warp=warps, image=s.image))
    args.k = k
    return cams, refs, srefs, (H, W, w_lr, h_lr), mine, total_refs


_pads = []


def sampled_mode_rate(args, dens, refs, dims, cfg, n_refs=16, cams_for_hot=None):
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
    # the pipeline's DEFAULT configuration (upstream_normaliser: the weights normalised with torch's own CPU f32 sum, upstream's library call):
    # reference i's aggregated map travels to the host on a side stream while the host sums reference i - 1's and launches its fused call
    # (core/hotpath.py::HotPath.begin_normaliser / finish_normaliser) - the launch stream never waits for the host
    from lichtfeld_densification_plugin_amd.core.hotpath import HotPath
    hot = HotPath(cams_for_hot, cfg, 0.9, wm, hm, dens.device, dens)
    passes_ms = []
    for warm in (True, False, False, False, False, False):      # one warm pass, five timed ones: the host side of this path (torch's CPU reduction on
        dens.seed_rng(cfg.seed)                                  # the host's thread pool) depends on what else runs on the host - the pool's four GPU slots share it
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts3, pend, fly = 0, [], []
        for r in todo:
            pend.append((r, hot.begin_normaliser(r, None)))
            while len(pend) > 1:
                r0, h0 = pend.pop(0)
                fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
            while len(fly) > 1:
                pts3 += getattr(hot.finish_sampled(fly.pop(0)), 'count', 0)
        while pend:
            r0, h0 = pend.pop(0)
            fly.append(hot.launch_sampled(r0, None, None, s_override=hot.finish_normaliser(h0), batch=h0[0]))
        while fly:
            pts3 += getattr(hot.finish_sampled(fly.pop(0)), 'count', 0)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        if not warm:
            passes_ms.append(dt3 / len(todo) * 1e3)
    res["default_config_ms_per_reference"] = min(passes_ms)
    res["default_config_points_per_reference"] = pts3 / len(todo)
    res["default_config_passes_ms_per_reference"] = passes_ms
    res["default_config_note"] = ("upstream_normaliser=True (the pipeline's default): upstream's torch f32 sum of every aggregated map on the host, the map copied on a "
                                  "side stream while the neighbouring references are launched / collected")
    # several references per fused call (lfd_triangulate_sampled_multi), every reference on its own MT19937 stream - what sharded
    # runs use (core/pipeline.py, per_reference_rng): one aggregate launch, the selections of the group side by side in one
    # launch (a selection occupies 17 of the 256 CUs), one pair of indexed launches, one read-back
    G = len(todo)
    outg = hb.OutputBuffers(cap * G, G, args.k, dens.device)
    bg = hb.PreparedBatch(todo, wm, hm)
    seeds = [(cfg.seed * 2654435761 + i) & 0xFFFFFFFF for i in range(G)]
    reps = 6
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptsg = 0
        for _ in range(reps):
            dens.launch_sampled_multi(bg, params, cfg.matches_per_ref, outg, seeds, cap=0.9, border=2, tiles=24)
            ptsg += outg.collect(indexed=True, check_selection=True).count
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t0
    res["grouped"] = {"references_per_call": G, "ms_per_reference": dtg / (reps * G) * 1e3, "refs_per_s": reps * G / dtg,
                      "pairs_per_s": reps * G * args.k / dtg, "points_per_s": ptsg / dtg,
                      "note": "per-reference MT19937 streams (sharded runs): the group's selections run side by side"}
    # round 5: the SAME single stream (upstream's semantics, bit for bit), several references per fused call: lfd_triangulate_sampled_chain - the
    # references draw one after the other from the context's stream, everything else of their selections runs side by side.  First the call alone
    # (the device's exact sums), then as the driver runs it by default (upstream's normaliser: every reference's weight map to the host, torch's sum)
    outc = [hb.OutputBuffers(cap * G, G, args.k, dens.device) for _ in range(2)]
    for warm in (True, False):
        dens.seed_rng(cfg.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptsc = 0
        dens.launch_sampled_chain(bg, params, cfg.matches_per_ref, outc[0], cap=0.9, border=2, tiles=24)
        for i in range(reps):
            if i + 1 < reps:
                dens.launch_sampled_chain(bg, params, cfg.matches_per_ref, outc[(i + 1) & 1], cap=0.9, border=2, tiles=24)
            ptsc += outc[i & 1].collect(indexed=True, check_selection=True).count
        torch.cuda.synchronize()
        dtc = time.perf_counter() - t0
    res["chained"] = {"references_per_call": G, "ms_per_reference": dtc / (reps * G) * 1e3, "refs_per_s": reps * G / dtc,
                      "pairs_per_s": reps * G * args.k / dtc, "points_per_s": ptsc / dtc,
                      "note": "ONE MT19937 stream (upstream's): the group's references draw in order, the rest of their selections side by side; device sums"}
    passes_c = []
    for warm in (True, False, False, False):
        dens.seed_rng(cfg.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptsd, ready, fly = 0, [], []
        for _ in range(reps):          # (core/strategies.py::SampledLoop._launch_chain: a group's weight maps travel while the group before it computes)
            b = bg                     # (the batch descriptor prebuilt, like the one-reference passes above)
            ready.append((b, hot.begin_chain_normalisers(b)))
            while len(ready) > 1:
                b0, s0 = ready.pop(0)
                fly.append(hot.launch_sampled_chain(b0, hot.finish_chain_normalisers(s0)))
            while len(fly) > 1:
                ptsd += getattr(hot.finish_sampled(fly.pop(0), check_selection=False), 'count', 0)
        while ready:
            b0, s0 = ready.pop(0)
            fly.append(hot.launch_sampled_chain(b0, hot.finish_chain_normalisers(s0)))
        while fly:
            ptsd += getattr(hot.finish_sampled(fly.pop(0), check_selection=False), 'count', 0)
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t0
        if not warm:
            passes_c.append(dtd / (reps * G) * 1e3)
    res["chained"]["default_config_ms_per_reference"] = min(passes_c)
    res["chained"]["default_config_passes_ms_per_reference"] = passes_c
    res["chained"]["default_config_points_per_reference"] = ptsd / (reps * G)
    return res


def unordered_rate(args, dens, batch, params, out, H, W, algo_bytes, launches=100):
    """The same kernel with UNORDERED retirement (lfd_triangulate_dense_segments, opt-in: no look-back, one atomic per tile, a tile table for the
    consumers that restore raster order): its own start / stop events over `launches` back-to-back launches of the headline batch."""
    n = batch.n_refs
    if out.capacity < n * H * W:
        return None
    tpr = (H * W + 1023) // 1024
    table = torch.zeros((n * tpr, 2), dtype=torch.int32, device=dens.device)
    counts = torch.zeros((n,), dtype=torch.int64, device=dens.device)
    for _ in range(16):
        dens.launch_dense_segments(batch, params, out, table, counts)
    dens.time_dense_kernels(launches)
    for _ in range(launches):
        dens.launch_dense_segments(batch, params, out, table, counts)
    ms = dens.dense_kernel_times_ms()
    dens.time_dense_kernels(0)
    dens.check_launches()
    k_ms = float(np.mean(ms))
    return {"kernel": "lfd_dense_segments_kernel", "kernel_ms": k_ms, "achieved": algo_bytes / (k_ms * 1e-3) / 1e9, "frac": algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "survivors": int(counts.sum().item()),
            "note": "opt-in (DensePipelineConfig.experimental['dense_tile_segments']): tiles claim output room with one atomic instead of the ordered look-back; lfd_order_segments / "
                    "lfd_pack_*_segments restore raster order from the tile table (bit-identical result, tests/test_gpu_segments.py; their cost: profiles/r4/ab_segments_v3_chunk_index.txt)"}


def ply_output_rate(args, dens, batch, params, H, W, cells, s_frac, launches=100):
    """The same kernel writing the file payload itself (lfd_triangulate_dense_ply: 15-byte PLY vertex records instead of the 28-byte arrays): its own
    start / stop events, against lfd_triangulate_dense + lfd_pack_ply - the pair it replaces for a consumer that writes or ships the PLY."""
    n = batch.n_refs
    rec = torch.empty((n * H * W * 15,), dtype=torch.uint8, device=dens.device)
    offs = torch.zeros((n + 1,), dtype=torch.int64, device=dens.device)
    for _ in range(16):
        dens.launch_dense_ply(batch, params, rec, offs)
    dens.time_dense_kernels(launches)
    for _ in range(launches):
        dens.launch_dense_ply(batch, params, rec, offs)
    ms = dens.dense_kernel_times_ms()
    dens.time_dense_kernels(0)
    dens.check_launches()
    k_ms = float(np.mean(ms))
    bytes_ply = cells * (4 * args.k + 11 + 15 * s_frac)
    # ... and without the ordered retirement (round 5: lfd_triangulate_dense_ply_segments, every reference's point set in its own region)
    tpr = dens.tiles_per_ref(H, W)
    table = torch.zeros((n * tpr, 2), dtype=torch.int32, device=dens.device)
    counts = torch.zeros((n,), dtype=torch.int64, device=dens.device)
    for _ in range(16):
        dens.launch_dense_ply_segments(batch, params, rec, counts, table)
    dens.time_dense_kernels(launches)
    for _ in range(launches):
        dens.launch_dense_ply_segments(batch, params, rec, counts, table)
    ms_u = dens.dense_kernel_times_ms()
    dens.time_dense_kernels(0)
    dens.check_launches()
    u_ms = float(np.mean(ms_u))
    unordered = {"kernel": "lfd_dense_ply_segments_kernel", "kernel_ms": u_ms, "achieved": bytes_ply / (u_ms * 1e-3) / 1e9, "frac": bytes_ply / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "survivors": int(counts.sum().item())}
    return {"kernel": "lfd_dense_ply_kernel", "kernel_ms": k_ms, "unordered": unordered, "bytes_per_cell": 4 * args.k + 11 + 15 * s_frac, "achieved": bytes_ply / (k_ms * 1e-3) / 1e9,
            "frac": bytes_ply / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "survivors": int(offs[-1].item()),
            "note": "15-byte records from the kernel itself: replaces lfd_triangulate_dense + lfd_pack_ply (secondary_kernels.lfd_pack_ply_kernel.ms) where the consumer "
                    "is the PLY writer / the exchange; bytes equal the packer's (tests/test_gpu_segments.py)"}


def device_copy_bandwidth(dev, n_bytes=1 << 30, reps=10):
    """HBM bytes moved per second by a device-to-device copy of 1 GiB (read + write counted), the practical ceiling beside the
    8 TB/s of the data sheet."""
    a = torch.empty(n_bytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    ms = _event_ms(lambda: b.copy_(a), reps)
    del a, b
    return 2.0 * n_bytes / (ms * 1e-3) / 1e9


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


def _oracle_cam(c):
    from oracle import densify_oracle as orc     # checker / baseline only
    return orc.OracleCamera(K=c.K, R=c.R, t=c.t, P=c.P, C=c.C, width=c.width, height=c.height)


def cpu_baseline(args, cams, srefs, dims, cfg):
    """CPU numbers for the same workload, on this box's host cores (bounded sample, ~10-20 s in all):
      * `value`: the CPU twin of the C-ABI (lfd_triangulate_dense_host: the HOST build of the kernels' per-cell source,
        csrc/lfd_geometry.hpp, on std::threads) over every cell of the sample, all host threads - the same mode as the
        GPU headline; `one_thread` the same on 1 thread;
      * `sampled_mode`: upstream's own mode (aggregate -> coverage sampling M=10000 -> triangulate the ~9k selected cells)
        on the twin + the host sampling stage, per reference - what upstream's `_triangulate_ref` does in 0.3-0.7 s;
      * `oracle_numpy`: the NumPy oracle (upstream's arithmetic incl. the batched LAPACK f32 SVD), 1 thread, dense;
      * `reference_python`: upstream's own Python timed in the development container (its files do
        not travel to the GPU box)."""
    if args.cpu_sample_refs <= 0:
        return None
    from lichtfeld_densification_plugin_amd.core.sampling import select_samples_with_coverage
    H, W, wm, hm = dims
    sample = srefs[:args.cpu_sample_refs]
    host_refs = [hb.ReferenceInputs(ref_cam=s.ref_index, nbr_cams=list(s.nbr_indices), cert=[s.cert[j].cpu() for j in range(args.k)],
                                    warp=[s.warp[j].cpu().contiguous() for j in range(args.k)], image=s.image.cpu()) for s in sample]
    batch = hb.PreparedBatch(host_refs, wm, hm, cameras=cams)
    params = hb.make_params(cfg)
    out = {}
    n_hw = os.cpu_count() or 1

    def timed_dense(twin, b, refs_used, min_s=2.0, max_passes=64):
        """Whole passes over the sample until at least ``min_s`` seconds have been measured (the pool's threads are parked between
        calls; the first, untimed pass faults the output pages in)."""
        twin.triangulate_dense(b, params)
        n, pts, t0 = 0, 0, time.perf_counter()
        while True:
            pts += twin.triangulate_dense(b, params).count
            n += 1
            dt = time.perf_counter() - t0
            if dt >= min_s or n >= max_passes:
                break
        return dict(points_per_s=pts / dt, cells_per_s=n * refs_used * H * W / dt, pairs_per_s=n * refs_used * args.k / dt,
                    seconds=dt, passes=n, references=refs_used, threads=twin.n_threads)

    ladder = sorted({t for t in (1, 16, 64, n_hw) if t <= n_hw})
    scaling = []
    for threads in ladder:
        twin = hb.HostDensifier(threads)
        twin.upload_cameras(cams)
        refs_used = len(host_refs) if threads >= 16 else max(1, len(host_refs) // 6)
        b = batch if refs_used == len(host_refs) else hb.PreparedBatch(host_refs[:refs_used], wm, hm, cameras=cams)
        r = timed_dense(twin, b, refs_used)
        scaling.append(r)
        if threads == 1:
            out["one"] = r
        if threads == n_hw:
            out["all"] = r
            rng = np.random.RandomState(cfg.seed)     # upstream's own mode on the twin, reference after reference
            t0 = time.perf_counter()
            pts = 0
            n_s = min(8, len(host_refs))
            for rf in host_refs[:n_s]:
                b1 = hb.PreparedBatch([rf], wm, hm, cameras=cams)
                best, _ = twin.aggregate(b1, params)
                sel = select_samples_with_coverage(best[0], cfg.matches_per_ref, cap=0.9, border=2, tiles=24, rng=rng)
                pts += twin.triangulate_indexed(b1, params, torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int64)), [0, int(sel.size)]).count
            dts = time.perf_counter() - t0
            out["sampled"] = dict(ms_per_reference=dts / n_s * 1e3, points_per_s=pts / dts, pairs_per_s=n_s * args.k / dts,
                                  references=n_s, matches_per_ref=cfg.matches_per_ref, threads=twin.n_threads,
                                  note="CPU twin (aggregate + indexed) + host coverage sampling (core/sampling.py)")
        twin.close()
    # the NumPy oracle on a few references (1 BLAS thread), as in round 1
    from oracle import densify_oracle as orc     # checker / baseline only
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                              # pragma: no cover
        threadpool_limits = None
    oparams = orc.OracleParams(certainty_thresh=cfg.certainty_thresh, reproj_thresh=cfg.reproj_thresh,
                               sampson_thresh=cfg.sampson_thresh, min_parallax_deg=cfg.min_parallax_deg)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    n_o = min(4, len(sample))
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    t0 = time.perf_counter()
    opts = 0
    with np.errstate(all="ignore"):
        for s in sample[:n_o]:
            opts += orc.triangulate_dense([s.cert[j].cpu().numpy() for j in range(args.k)], [s.warp[j].cpu().numpy() for j in range(args.k)],
                                          s.image.cpu().numpy(), _oracle_cam(cams[s.ref_index]), [_oracle_cam(cams[n]) for n in s.nbr_indices],
                                          wm, hm, oparams, axes=axes)["xyz"].shape[0]
    dto = time.perf_counter() - t0
    if ctx is not None and hasattr(ctx, "restore_original_limits"):
        ctx.restore_original_limits()
    a = max(scaling, key=lambda r: r["points_per_s"])         # the fastest rung of the ladder: more threads than the box really gives are slower
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            cpu_max = fh.read().strip()
    except OSError:
        cpu_max = None
    return {"value": a["points_per_s"], "unit": "points/s", "cores": a["threads"], "kind": "port", "host_cpus": n_hw, "cgroup_cpu_max": cpu_max,
            "sample": f"{a['passes']} passes over {a['references']} reference views x {args.k} neighbours x {H}x{W} cells of this workload, dense mode "
                      f"({a['seconds']:.2f} s on {a['threads']} threads of {n_hw} host CPUs; CPU twin of the C-ABI = host build of the kernels' source, "
                      f"persistent thread pool)",
            "scaling": [{"threads": r["threads"], "points_per_s": r["points_per_s"], "seconds": r["seconds"], "passes": r["passes"],
                         "references": r["references"]} for r in scaling],
            "pairs_per_s": a["pairs_per_s"], "cells_per_s": a["cells_per_s"],
            "one_thread": {"value": out["one"]["points_per_s"], "cores": 1, "pairs_per_s": out["one"]["pairs_per_s"],
                           "sample": f"{out['one']['passes']} passes over {out['one']['references']} references, {out['one']['seconds']:.2f} s"},
            "sampled_mode": out["sampled"],
            "oracle_numpy": {"value": opts / dto, "cores": 1, "pairs_per_s": n_o * args.k / dto,
                             "sample": f"{n_o} references dense, {dto:.1f} s, BLAS threads limited to 1 (upstream's arithmetic: batched LAPACK f32 SVD)"},
            "reference_python": reference_python_record()}


def reference_python_record():
    """Upstream's OWN Python on this benchmark's workloads, timed in the development container by tests/golden/time_reference.py (which imports
    /root/reference; its files cannot travel to the GPU box) and committed as tests/golden/g11_reference_timing.json: `_triangulate_ref` on references
    of the headline workload, `run_dense_pipeline` + `write_ply` on the `pipeline` leg's on-disk scene."""
    path = os.path.join(ROOT, "tests", "golden", "g11_reference_timing.json")
    try:
        with open(path) as fh:
            g = json.load(fh)
    except Exception as exc:
        return {"note": f"tests/golden/g11_reference_timing.json not readable ({exc})"}
    tri = g["triangulate_ref"]["gui_k3"]
    pipe = g["run_dense_pipeline"]
    return {"source": "tests/golden/g11_reference_timing.json (tests/golden/time_reference.py, development container; not runnable on the GPU box)",
            "cores": g["host"]["cores_usable"], "torch_threads": g["host"]["torch_threads"], "versions": g["versions"],
            "triangulate_ref": {name: {"ms_per_reference": r["seconds_per_reference"]["mean"] * 1e3, "points_per_s": r["points_per_s"], "pairs_per_s": r["pairs_per_s"],
                                       "points_per_reference": r["points_per_reference"], "references": r["references"], "neighbours": r["neighbours"],
                                       "matches_per_ref": r["matches_per_ref"]} for name, r in g["triangulate_ref"].items()},
            "ms_per_reference": tri["seconds_per_reference"]["mean"] * 1e3, "points_per_s": tri["points_per_s"],
            "run_dense_pipeline": {name: {k_: r[k_] for k_ in ("pack_workers", "cameras", "references", "pairs", "points", "seconds", "matcher_seconds", "refs_per_s",
                                                              "pairs_per_s", "points_per_s", "write_ply_seconds", "viz_interval", "previews", "preview_bytes") if k_ in r}
                                   for name, r in pipe.items()},
            "note": "upstream core/pipeline.py::_triangulate_ref (sampled mode, 512^2, GUI k = 3 / M = 10000 and CLI k = 4 / M = 12000) on references of this "
                    "workload, and upstream's run_dense_pipeline + write_ply on the pipeline leg's scene (same images, plan and matcher fields; "
                    "`pack_workers_4_previews_every_3`: with the GUI's intermediate previews, what the pipeline leg's `gui.previews_every_3` runs)"}


def parity_report(args, dens, cams, refs, srefs, dims, cfg):
    """"the bench reports near-threshold counts separately".  The first --parity-refs references of the
    timed workload through the dense kernel (with cell indices) against the oracle, every differing cell classified by
    the threshold whose derived rounding band explains it (oracle.classify_flips); `flipped_out_of_band` must be 0."""
    if args.parity_refs <= 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import flip_report          # tests/helpers.py: oracle-based checker
    from oracle import densify_oracle as orc
    H, W, wm, hm = dims
    n = min(args.parity_refs, len(refs))
    batch = hb.PreparedBatch(refs[:n], wm, hm, cameras=cams)
    out = dens.triangulate_dense(batch, hb.make_params(cfg))
    cell = out.cell.cpu().numpy().astype(np.int64)
    oparams = orc.OracleParams(certainty_thresh=cfg.certainty_thresh, reproj_thresh=cfg.reproj_thresh,
                               sampson_thresh=cfg.sampson_thresh, min_parallax_deg=cfg.min_parallax_deg)
    axes = (orc.identity_axis_scalar(W), orc.identity_axis_scalar(H))
    tot = {"cells": 0, "flipped": 0, "flipped_out_of_band": 0, "by_reason": {r: 0 for r in orc.FLIP_REASONS}}
    for r in range(n):
        lo, hi = int(out.ref_offsets[r]), int(out.ref_offsets[r + 1])
        rep = flip_report(cell[lo:hi], srefs[r], cams, wm, hm, oparams, axes)
        tot["cells"] += rep["cells"]; tot["flipped"] += rep["flipped"]; tot["flipped_out_of_band"] += rep["out_of_band"]
        for k_, v in rep["by_reason"].items():
            tot["by_reason"][k_] += v
    tot["references"] = n
    tot["survivors_kernel"] = int(out.count)
    tot["note"] = ("cells kept by exactly one of {HIP dense kernel, oracle = upstream's f32 arithmetic}; in band = the threshold lies between "
                   "upstream's f32 value and the rounding-free value of the same formula (+ one f32 evaluation's bound)")
    return tot



class _Pieces:
    """A rank's own records as the list of buffers they sit in (nothing concatenated); numel() is all the bench needs of it."""

    def __init__(self, parts):
        self.parts = parts

    def numel(self):
        return sum(int(p.numel()) for p in self.parts)


def run_sharded(args, world, rank, dev, dist, backend):
    """--gpus N > 1 (or one rank with LFD_BENCH_FORCE_DIST=1): ONE scene over the ranks, and the exchange of the survivors INSIDE the timed region.
    The scene's references are split (core/distributed.py::plan_replication, from THIS run's own measurements) into a sharded prefix - dealt
    round-robin, each rank's share in ``--exchange-rounds`` launches of the fused dense kernel, every launch's survivors (15-byte PLY records written
    by the kernel itself, lfd_triangulate_dense_ply; or the 28-byte rows) handed on a side stream to the round's asynchronous collective
    (core/distributed.py::OverlappedExchange) - and a replicated suffix that every rank that receives the cloud computes itself while the rounds
    travel: one GPU triangulates a reference faster than its survivors cross an xGMI link, so recomputing part of the scene is cheaper than moving
    it.  When a step ends the ordered cloud of the whole scene is one contiguous buffer on every rank (all_gather) / on rank 0 (gather_to_root).
    `value` = the scene's survivors, each counted ONCE, / that time.  Beside it: the pure-sharding schedule (`value_pure_sharding`), the same steps
    without any exchange, with the counts only, the end-of-run exchanges of round 3, and the upstream-equivalent sampled mode through the exchange."""
    from lichtfeld_densification_plugin_amd.core import distributed as lfd_dist
    h_lr, w_lr, H, W = synthetic.ROMA_PRESETS[args.preset]
    wm, hm = w_lr, h_lr
    total_refs = args.refs * world if args.scaling == "weak" else args.refs
    if total_refs < world:
        raise SystemExit(f"--refs {args.refs} ({args.scaling} scaling) leaves a rank without a reference view: {total_refs} references over {world} ranks")
    ply = args.exchange_records == "ply"
    rec_bytes, cols, rdtype = (15, 15, torch.uint8) if ply else (28, 7, torch.float32)
    cdev = dev if backend != "gloo" else torch.device("cpu")
    consumes_cloud = args.exchange == "all_gather" or rank == 0
    side = torch.cuda.Stream(device=dev)
    cache = {}                                     # global position -> ReferenceInputs (generated once per rank)
    cams_box = []

    def refs_at(positions):
        need = [g for g in positions if g not in cache]
        if need or not cams_box:
            cams, refs, _srefs, _dims, mine, _tot = build_workload(args, rank, world, dev, positions=need)
            if not cams_box:
                cams_box.append(cams)
            for g, r in zip(mine, refs):
                cache[g] = r
        return [cache[g] for g in positions]

    def barrier():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # ranks the communicator actually reports: an all-reduce of ones ON THE DEVICE through it (0 when the backend is not RCCL)
    ones = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    rccl = int(ones.item()) if backend == "nccl" else 0
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)
    refs_at(lfd_dist.shard_references(total_refs, rank, world))          # (also fixes args.k and the camera ring)
    cams = cams_box[0]
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)
    params = hb.make_params(cfg)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)

    class Schedule:
        """One way of running the scene: the last `n_rep` references replicated, the others sharded in rounds."""

        def __init__(self, n_rep, rounds=None):
            self.n_rep = int(n_rep)
            rounds = int(rounds) if rounds else int(args.exchange_rounds)
            mine, self.n_sh = lfd_dist.split_replicated(total_refs, self.n_rep, rank, world, replicas_here=consumes_cloud)
            self.sh_pos = [g for g in mine if g < self.n_sh]
            self.rep_pos = [g for g in mine if g >= self.n_sh]
            n_local_max = (self.n_sh + world - 1) // world
            self.per_round = max(1, -(-n_local_max // max(1, rounds))) if self.n_sh else 1
            self.n_rounds = -(-n_local_max // self.per_round) if self.n_sh else 0
            sh_refs = refs_at(self.sh_pos)
            self.chunks = [sh_refs[c * self.per_round:(c + 1) * self.per_round] for c in range(self.n_rounds)]
            self.batches = [hb.PreparedBatch(ch, wm, hm, cameras=cams) if ch else None for ch in self.chunks]
            self.rep_batch = hb.PreparedBatch(refs_at(self.rep_pos), wm, hm, cameras=cams) if self.rep_pos else None
            # "ply": the kernel writes the 15-byte records itself (lfd_triangulate_dense_ply - no packing pass); "f32": the 28-byte arrays, made into rows
            self.outs = [hb.OutputBuffers(len(ch) * H * W, len(ch), args.k, dev, with_cell=False, with_segments=False) if (ch and not ply) else None for ch in self.chunks]
            self.recs_dev = [torch.empty((len(ch) * H * W * 15,), dtype=torch.uint8, device=dev) if (ch and ply) else None for ch in self.chunks]
            self.offs_dev = [torch.zeros((len(ch) + 1,), dtype=torch.int64, device=dev) if ch else None for ch in self.chunks]
            self.offs_host = [torch.zeros((len(ch) + 1,), dtype=torch.int64).pin_memory() if ch else None for ch in self.chunks]
            self.done = [torch.cuda.Event() if ch else None for ch in self.chunks]
            # the cloud of a rank that consumes it: ONE buffer; the sharded part ends at row `cap_sh` (the exchange places its ordered records
            # there, finish(place=...)), the replicated part starts there (the kernel writes it in place): contiguous, in global reference order
            self.cap_sh = self.n_sh * H * W
            self.cloud = None
            if consumes_cloud and (self.rep_pos or self.n_sh):
                self.cloud = torch.empty(((self.cap_sh + len(self.rep_pos) * H * W) * cols,), dtype=rdtype, device=dev)
            self.rep_out = (hb.OutputBuffers(len(self.rep_pos) * H * W, len(self.rep_pos), args.k, dev, with_cell=False, with_segments=False)
                            if (self.rep_pos and not ply) else None)
            self.rep_offs_dev = torch.zeros((len(self.rep_pos) + 1,), dtype=torch.int64, device=dev) if self.rep_pos else None
            self.rep_offs_host = torch.zeros((len(self.rep_pos) + 1,), dtype=torch.int64).pin_memory() if self.rep_pos else None
            self.launches_per_step = sum(1 for b in self.batches if b is not None) + (1 if self.rep_batch is not None else 0)

        def launch(self, c):
            if ply:
                dens.launch_dense_ply(self.batches[c], params, self.recs_dev[c], self.offs_dev[c])
            else:
                dens.launch_dense(self.batches[c], params, self.outs[c])
                self.offs_dev[c].copy_(self.outs[c].ref_offsets)
            with torch.cuda.stream(dens.stream):
                self.offs_host[c].copy_(self.offs_dev[c], non_blocking=True)       # the counts of this launch, behind it on the launch stream
                self.done[c].record(dens.stream)

        def launch_replicated(self):
            """the replicated references, in place: records straight into the cloud buffer behind the sharded part"""
            if ply:
                dens.launch_dense_ply(self.rep_batch, params, self.cloud[self.cap_sh * 15:], self.rep_offs_dev)
            else:
                dens.launch_dense(self.rep_batch, params, self.rep_out)
                self.rep_offs_dev.copy_(self.rep_out.ref_offsets)
            with torch.cuda.stream(dens.stream):
                self.rep_offs_host.copy_(self.rep_offs_dev, non_blocking=True)

        def retire(self, c, ex):
            """launch c is through (its counts behind their own event): hand every reference's records to the round, on the side stream"""
            if self.batches[c] is None:
                return 0
            self.done[c].synchronize()
            offs = self.offs_host[c].numpy().copy()
            if ex is not None:
                with torch.cuda.stream(side):
                    side.wait_event(self.done[c])
                    o = self.outs[c]
                    body = self.recs_dev[c] if ply else lfd_dist.rows_from_points(o.xyz[:int(offs[-1])], o.rgb[:int(offs[-1])], o.err[:int(offs[-1])])
                    ex.push_many(c * self.per_round, body, np.diff(offs))          # views of the launch's own buffer: nothing is copied
            return int(offs[-1])

        def scene(self, with_exchange=True, form=None):
            """-> (this rank's sharded survivors, replicated survivors, the cloud (or this rank's shard), global counts of the sharded part)"""
            the_form = form or args.exchange
            on_device = cdev == dev                          # RCCL: the records never leave HBM; gloo (ranks sharing a GPU): through host tensors
            # nothing replicated: the ordered records are placed from row 0 of the cloud as the rounds complete; with a replicated part behind
            # them their place is only known at the end (finish(place=...)): they must END where the replicated part begins
            early = with_exchange and consumes_cloud and on_device and self.rep_batch is None and the_form != "counts_only" and self.cloud is not None
            ex = (lfd_dist.OverlappedExchange(dist, self.n_sh, self.per_round, dev, form=the_form, record="ply" if ply else "f32", eager=True,
                                              dest=self.cloud.view(-1, cols) if early else None)
                  if (with_exchange and self.n_sh) else None)
            pts = 0
            for c in range(self.n_rounds):
                if self.batches[c] is not None:
                    self.launch(c)
                if c >= 1:
                    pts += self.retire(c - 1, ex)
            if self.rep_batch is not None:
                self.launch_replicated()                     # computes while the rounds travel
            if self.n_rounds:
                pts += self.retire(self.n_rounds - 1, ex)
            n_rep_pts = 0
            if self.rep_batch is not None:
                dens.stream.synchronize()
                n_rep_pts = int(self.rep_offs_host[-1])
                if not ply:                                  # the rows of the replicated part, behind the sharded part
                    with torch.cuda.stream(dens.stream):
                        dst = self.cloud.view(-1, 7)[self.cap_sh:self.cap_sh + n_rep_pts]
                        dst[:, 0:3] = self.rep_out.xyz[:n_rep_pts]; dst[:, 3:6] = self.rep_out.rgb[:n_rep_pts]; dst[:, 6] = self.rep_out.err[:n_rep_pts]
            if ex is None:
                if with_exchange and self.rep_batch is not None:        # everything replicated: the cloud is what this rank computed, nothing travels
                    dens.stream.synchronize()
                    return pts, n_rep_pts, self.cloud[self.cap_sh * cols:(self.cap_sh + n_rep_pts) * cols], np.zeros((0,), np.int64)
                return pts, n_rep_pts, None, None
            placed = {}

            def place(n_rows):
                placed["n"] = n_rows
                return self.cloud.view(-1, cols)[self.cap_sh - n_rows:self.cap_sh]
            in_place = consumes_cloud and on_device and the_form != "counts_only" and not early
            with torch.cuda.stream(side):
                recs, counts = ex.finish(place=place if in_place else None, concat=False)
            if isinstance(recs, list):                       # this rank's own shard, where the launches left it
                recs = _Pieces(recs)
            side.synchronize()
            if self.rep_batch is not None:
                dens.stream.synchronize()
            if in_place:                                     # sharded part | replicated part: one contiguous slice, global reference order
                m_sh = placed.get("n", 0)
                recs = self.cloud[(self.cap_sh - m_sh) * cols:(self.cap_sh + n_rep_pts) * cols]
            elif consumes_cloud and self.rep_batch is not None and (form or args.exchange) != "counts_only":      # (gloo: the records came back through the host)
                recs = torch.cat([recs.reshape(-1), self.cloud[self.cap_sh * cols:(self.cap_sh + n_rep_pts) * cols]])
            return pts, n_rep_pts, recs, counts

    def timed(sched, steps, **kw):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = sched.scene(**kw)
        barrier()
        return time.perf_counter() - t0, r

    # ---- 1. the pure-sharding schedule: spin-up, warm-up (also creates the communicator's channels), K steps -----------------------------------
    pure = Schedule(0)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:
        pure.scene(with_exchange=False)
    for _ in range(max(args.warmup, 1)):
        pure.scene()
    dens.time_dense_kernels(args.steps * pure.launches_per_step)
    if os.environ.get("LFD_BENCH_CPROFILE"):          # where the host's time of a step goes (profiles/r4/exchange_overhead.txt)
        import cProfile
        import pstats
        for label, kw in (("all_gather", {}), ("counts_only", {"form": "counts_only"}), ("no exchange", {"with_exchange": False})):
            pr = cProfile.Profile()
            pr.enable()
            el_p, _r = timed(pure, args.steps, **kw)
            pr.disable()
            print(f"==== {label}: {el_p / args.steps * 1e3:.3f} ms per step ====", file=sys.stderr)
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(22)
    el_pure, (n_pts_pure, _z, recs, counts) = timed(pure, args.steps)
    per_launch = [float(x) for x in dens.dense_kernel_times_ms()]
    dens.time_dense_kernels(0)
    kernel_ms_step = float(np.sum(per_launch) / max(args.steps, 1))
    dens.check_launches()
    total_pts = int(counts.sum())
    if consumes_cloud:
        assert recs.numel() == total_pts * cols, (recs.shape, total_pts)
    el_compute, _r = timed(pure, args.steps, with_exchange=False)                    # the same K steps without any exchange
    # ... and with the counts only on the wire: the cloud stays sharded over the ranks' HBM, every rank knows where its references sit in the
    # 1-GPU sequence (what a data-parallel consumer or the byte-range writer of core/distributed.py::SharedFilePlyStream needs)
    pure.scene(form="counts_only")
    el_resident, (_n, _z, own_recs, counts_co) = timed(pure, args.steps, form="counts_only")
    assert int(counts_co.sum()) == total_pts and own_recs.numel() == n_pts_pure * cols
    el_pure, el_compute, kernel_ms_max, el_resident = max_over_ranks([el_pure, el_compute, kernel_ms_step, el_resident])

    # ---- 2. what this machine does per reference and per byte: the plan's inputs, measured here ------------------------------------------------
    n_loc = len(pure.sh_pos)
    # one launch of n references = launch_ms + ref_ms * n: two points of the line, back-to-back launches of the whole share and of half of it
    def launch_time(n):
        b = hb.PreparedBatch(refs_at(pure.sh_pos[:n]), wm, hm, cameras=cams)
        rec = torch.empty((n * H * W * 15,), dtype=torch.uint8, device=dev) if ply else None
        o = hb.OutputBuffers(n * H * W, n, args.k, dev, with_cell=False, with_segments=False) if not ply else None
        off = torch.zeros((n + 1,), dtype=torch.int64, device=dev)

        def go():
            if ply:
                dens.launch_dense_ply(b, params, rec, off)
            else:
                dens.launch_dense(b, params, o)
        for _ in range(3):
            go()
        torch.cuda.synchronize(dev)
        t_0 = time.perf_counter()
        reps_l = max(5, min(args.steps, 50))
        for _ in range(reps_l):
            go()
        torch.cuda.synchronize(dev)
        dens.check_launches()
        return (time.perf_counter() - t_0) / reps_l * 1e3
    n_hi, n_lo = n_loc, max(1, n_loc // 2)
    t_hi = launch_time(n_hi)
    if n_hi > n_lo:
        t_lo = launch_time(n_lo)
        ref_ms = max((t_hi - t_lo) / (n_hi - n_lo), 1e-6)
        launch_ms = max(t_lo - ref_ms * n_lo, 0.0)
    else:
        ref_ms, launch_ms = t_hi / max(n_hi, 1), 0.0
    ref_bytes = rec_bytes * total_pts / max(total_refs, 1)
    # the all-gather of one rank's share of the records as this backend moves it: per-peer bandwidth and the fixed cost
    shard_rows = max(1, int(total_pts / world))
    ag_in = torch.zeros((shard_rows * cols,), dtype=rdtype, device=cdev)
    ag_out = torch.empty((world * shard_rows * cols,), dtype=rdtype, device=cdev)
    small_in = torch.zeros((cols,), dtype=rdtype, device=cdev)
    small_out = torch.empty((world * cols,), dtype=rdtype, device=cdev)
    t_ag = {}
    for name, (i_, o_) in (("big", (ag_in, ag_out)), ("small", (small_in, small_out))):
        dist.all_gather_into_tensor(o_, i_)
        barrier()
        ta = time.perf_counter()
        for _ in range(3):
            dist.all_gather_into_tensor(o_, i_)
        barrier()
        t_ag[name] = (time.perf_counter() - ta) / 3 * 1e3
    del ag_in, ag_out
    t_big, t_small = max_over_ranks([t_ag["big"], t_ag["small"]])
    link_gbps = shard_rows * rec_bytes / max(t_big - t_small, 1e-3) / 1e6 if world > 1 else 122.0
    cp_src = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
    cp_dst = torch.empty_like(cp_src)
    cp_dst.copy_(cp_src)
    torch.cuda.synchronize(dev)
    tc0 = time.perf_counter()
    for _ in range(5):
        cp_dst.copy_(cp_src)
    torch.cuda.synchronize(dev)
    copy_gbps = 5 * (64 << 20) / (time.perf_counter() - tc0) / 1e9
    del cp_src, cp_dst
    ref_ms, launch_ms, link_inv, copy_inv, coll_ms = max_over_ranks([ref_ms, launch_ms, 1.0 / link_gbps, 1.0 / copy_gbps, t_small])
    plan = lfd_dist.plan_replication(total_refs, world, ref_ms, ref_bytes, launch_ms=launch_ms, link_gbps=1.0 / link_inv, collective_ms=2 * coll_ms,
                                     copy_gbps=1.0 / copy_inv)
    if args.replicate != "auto":
        plan = dict(plan, n_replicated=max(0, min(int(args.replicate), total_refs)), forced=True)
        plan["n_sharded"] = total_refs - plan["n_replicated"]
    elif world == 1:
        plan = dict(plan, n_replicated=0, n_sharded=total_refs)
    # The model knows the link and the kernel, not the software around them: `auto` MEASURES the candidates - the model's pick, nothing / a quarter /
    # half / three quarters / everything replicated - a few steps each, and takes the fastest (pure sharding and "every rank computes everything"
    # are always among them: the planned schedule is never slower than either).  The ranks agree on the times (MAX), hence on the choice.
    # (a round costs a chain of two collectives and a look at the counts - ~0.2 ms through RCCL, profiles/r4/exchange_overhead.txt - so the
    # candidates with a replicated part, whose compute is what the exchange hides behind, run ONE round; pure sharding is tried both ways)
    measured = {(0, pure.n_rounds): el_pure / args.steps * 1e3}
    best_rounds = pure.n_rounds
    if args.replicate == "auto" and world > 1:
        probe_steps = max(4, min(args.steps, 15))
        cands = [(0, 1)] if pure.n_rounds != 1 else []
        # (never "everything replicated": every rank would compute the whole scene, nothing would cross a link, and the headline would be the 1-GPU
        # number - not a measurement of strong scaling; at least one reference per rank stays sharded)
        most = max(0, total_refs - world)
        cands += [(c_, 1) for c_ in sorted({min(int(plan["n_replicated"]), most), total_refs // 4, total_refs // 2, min((3 * total_refs) // 4, most)} - {0})]
        for cand, rnd in cands:
            trial = Schedule(cand, rnd)
            trial.scene()
            el_c, _r = timed(trial, probe_steps)
            (el_c,) = max_over_ranks([el_c])
            measured[(cand, trial.n_rounds)] = el_c / probe_steps * 1e3
            del trial
            torch.cuda.empty_cache()
        best, best_rounds = min(measured, key=lambda c_: (measured[c_], c_))
        plan = dict(plan, model_n_replicated=int(plan["n_replicated"]), n_replicated=int(best), n_sharded=total_refs - int(best))

    # ---- 3. the planned schedule: `value` ----------------------------------------------------------------------------------------------------------
    n_rep = int(plan["n_replicated"])
    if n_rep > 0 or best_rounds != pure.n_rounds:
        sched = Schedule(n_rep, best_rounds if args.replicate == "auto" else None)
        for _ in range(max(args.warmup, 1)):
            sched.scene()
        el_value, (n_sh_pts, n_rep_pts, recs_v, counts_v) = timed(sched, args.steps)
        (el_value,) = max_over_ranks([el_value])
        sh_total = int(counts_v.sum()) if counts_v is not None else 0
        rep_t = torch.tensor([n_rep_pts], dtype=torch.int64, device=cdev)
        dist.all_reduce(rep_t, op=dist.ReduceOp.MAX)                 # (a rank that does not consume the cloud computed no replicated part)
        assert sh_total + int(rep_t.item()) == total_pts, (sh_total, int(rep_t.item()), total_pts)          # the same scene, every point once
        if consumes_cloud:
            assert recs_v.numel() == total_pts * cols, (recs_v.numel(), total_pts)
            if recs is not None and recs.numel() == recs_v.numel():
                assert torch.equal(recs_v.reshape(-1).to(recs.device), recs.reshape(-1)), "the planned schedule's cloud differs from the pure-sharding one"
    else:
        sched, el_value = pure, el_pure

    # round 3's end-of-run exchanges (28-byte rows, one collective after the last reference) on the pure-sharding survivors, for comparison
    end_of_run = {}
    outs = [hb.OutputBuffers(len(ch) * H * W, len(ch), args.k, dev, with_cell=False, with_segments=False) if ch else None for ch in pure.chunks]
    for c in range(pure.n_rounds):
        if pure.batches[c] is not None:
            dens.launch_dense(pure.batches[c], params, outs[c])
    res_all = [outs[c].collect() for c in range(pure.n_rounds) if pure.batches[c] is not None]
    lx = torch.cat([r.xyz for r in res_all]) if res_all else torch.zeros((0, 3), device=dev)
    lc = torch.cat([r.rgb for r in res_all]) if res_all else torch.zeros((0, 3), device=dev)
    le = torch.cat([r.err for r in res_all]) if res_all else torch.zeros((0,), device=dev)
    counts_local = [int(r.ref_offsets[i + 1] - r.ref_offsets[i]) for r in res_all for i in range(len(r.ref_offsets) - 1)]
    for name, fn in (("allgather_ms", lfd_dist.all_gather_by_reference), ("gather_to_root_ms", lfd_dist.gather_to_root_by_reference)):
        fn(lx, lc, le, counts_local, total_refs, dist)
        barrier()
        t_e = time.perf_counter()
        g = fn(lx, lc, le, counts_local, total_refs, dist)
        barrier()
        (dt_e,) = max_over_ranks([time.perf_counter() - t_e])
        end_of_run[name] = dt_e * 1e3
        assert int(g[3].sum()) == total_pts
    del outs, res_all, lx, lc, le

    # the upstream-equivalent mode through the same exchange: every reference's ~9.1k selected cells (M = 10000), per-reference RNG streams,
    # the rank's share in ONE fused call, f32 rows in one round
    sampled = None
    if not args.light or world > 1:
        M = cfg.matches_per_ref
        capn = M + 24 * 24 + 64
        my_refs = refs_at(pure.sh_pos)
        G = len(my_refs)
        bg = hb.PreparedBatch(my_refs, wm, hm, cameras=cams)
        outg = hb.OutputBuffers(capn * G, G, args.k, dev)
        seeds = [(cfg.seed * 2654435761 + gi) & 0xFFFFFFFF for gi in pure.sh_pos]
        n_local_max = (total_refs + world - 1) // world

        def sampled_scene():
            ex = lfd_dist.OverlappedExchange(dist, total_refs, n_local_max, dev, form=args.exchange, record="f32")
            dens.launch_sampled_multi(bg, params, M, outg, seeds, cap=0.9, border=2, tiles=24)
            r = outg.collect(indexed=True, check_selection=True)
            rows = lfd_dist.rows_from_points(r.xyz, r.rgb, r.err)
            for i in range(G):
                lo, hi = int(r.ref_offsets[i]), int(r.ref_offsets[i + 1])
                ex.push(i, rows[lo:hi] if hi > lo else None)
            return ex.finish()
        sampled_scene()
        barrier()
        ts = time.perf_counter()
        reps = max(3, min(args.steps, 20))
        for _ in range(reps):
            srecs, scounts = sampled_scene()
        barrier()
        (dts,) = max_over_ranks([time.perf_counter() - ts])
        sampled = {"value": float(scounts.sum()) * reps / dts, "unit": "points/s", "ms_per_scene": dts / reps * 1e3, "points_per_scene": int(scounts.sum()),
                   "pairs_per_s": total_refs * args.k * reps / dts, "matches_per_ref": M,
                   "note": "upstream-equivalent mode (coverage sampling, ~9.1k cells per reference) on the same scene and shards, 28-byte rows through "
                           "the same collective in one round, INSIDE the time (never replicated: its exchange is small)"}

    if rank == 0:
        n_rank_refs = len(pure.sh_pos)
        cells_rank = n_rank_refs * H * W
        s_frac = (n_pts_pure / cells_rank) if cells_rank else 0.0
        bytes_per_cell = 4 * args.k + 11 + (15 if ply else 28) * s_frac        # what this launch moves at least: k certainties, the winner's warp, a texel; 15 / 28 B per survivor
        kname = "lfd_dense_ply_kernel" if ply else "lfd_dense_kernel"
        step_ms = el_value / args.steps * 1e3
        n_sh = int(plan["n_sharded"])
        line = {
            "metric": "triangulated points/sec + pairs/sec, MipNeRF360 garden @fast, 1/2/4/8 GPU",
            "value": total_pts * args.steps / el_value, "unit": "points/s", "n_gpus": world, "rccl_ranks": rccl, "collective_backend": backend,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32 (+f64 Sampson/DLT solve)", "data": "synthetic",
            "value_includes": (f"every point of the scene ONCE: compute + the {args.exchange} of the survivors ({rec_bytes}-byte records) of {n_sh} sharded "
                               f"references in {sched.n_rounds} round(s) beside the compute; {n_rep} references replicated (computed by every rank that "
                               "receives the cloud instead of travelling); the ordered cloud ends up contiguous where the collective delivers it"),
            "config": {"workload": f"{WORKLOADS[args.workload]['what']}: ring of {WORKLOADS[args.workload]['n_cams']} cameras {WORKLOADS[args.workload].get('width', 1297)}x{WORKLOADS[args.workload].get('height', 840)}, `{args.preset}` grid {H}x{W}, "
                                   f"{args.refs} reference views x {args.k} neighbours {'per GPU' if args.scaling == 'weak' else 'in total, dealt over the ranks'}, "
                                   f"default thresholds (certainty 0.2 / sampson 5.0 / reproj 0.8 / parallax 0.5 deg), noise {args.noise_px} px, {args.outliers:.0%} outliers",
                       "kernel": f"fused dense filter+triangulate kernel ({kname})", "mode": "dense", "refs_per_gpu": n_rank_refs, "refs_total": total_refs,
                       "neighbours": args.k, "grid": [H, W], "sharding": f"{n_sh} references round-robin over {world} rank(s), {n_rep} replicated",
                       "launches_per_step": sched.launches_per_step, "refs_per_round": sched.per_round},
            "pairs_per_s": total_refs * args.k * args.steps / el_value,
            "cells_per_s": total_refs * H * W * args.steps / el_value,
            "survivor_fraction": s_frac,
            "replication": {"n_replicated": n_rep, "n_sharded": n_sh, "planned_step_ms": plan["step_ms"], "planned_pure_sharding_ms": plan["pure_sharding_ms"],
                            "planned_single_rank_ms": plan["single_rank_ms"], "inputs_measured_in_this_run": plan["inputs"], "forced": bool(plan.get("forced")),
                            "model_n_replicated": plan.get("model_n_replicated"), "candidates_measured_ms": {f"{k_[0]} replicated, {k_[1]} round(s)": v_ for k_, v_ in sorted(measured.items())},
                            "redundant_cells_per_step": n_rep * H * W * (world - 1 if args.exchange == "all_gather" else 0),
                            "note": "core/distributed.py::plan_replication: the references whose recomputation on every rank is cheaper than their "
                                    "survivors' trip over a link; with a matcher in the loop (tens of ms per pair) the plan is 0"},
            "nothing_sharded": n_sh == 0,     # a forced --replicate of the whole scene: `value` is then one GPU's work N times over, not strong scaling
            "value_pure_sharding": total_pts * args.steps / el_pure, "pure_sharding_ms": el_pure / args.steps * 1e3,
            "value_compute_only": total_pts * args.steps / el_compute, "compute_ms": el_compute / args.steps * 1e3,
            "kernel_ms": kernel_ms_max, "exchange_ms_exposed": (el_pure - el_compute) / args.steps * 1e3,
            "value_sharded_resident": total_pts * args.steps / el_resident,
            "value_sharded_resident_note": "the pure-sharding steps with only the per-reference counts exchanged (form counts_only): the ordered cloud stays sharded over the "
                                           "ranks' HBM with every reference's global offset known on every rank - comparable to N = 1, whose result also stays in HBM",
            "host": getattr(args, "host", None),
            "end_to_end": None,
            "end_to_end_note": "points/s including the RoMa-v2 forward (SURVEY 8d iii) is unmeasured: neither the RoMa-v2 weights nor torchvision are on the box",
            "exchange": {"form": args.exchange, "record_bytes": rec_bytes, "rounds": sched.n_rounds, "overlapped": True, "points": total_pts,
                         "bytes_per_rank_sent": int(rec_bytes * n_pts_pure), "bytes_gathered": int(rec_bytes * total_pts), "backend": backend,
                         "order": "global reference position (1-GPU sequence)", "end_of_run_28B": end_of_run,
                         "allgather_GBps_per_peer_measured": 1.0 / link_inv, "collective_ms_measured": coll_ms, "note": "bytes_*: the pure-sharding schedule"},
            "roofline": {"bound": "hbm", "kernel": kname, "kernel_ms": kernel_ms_step,
                         "achieved": (cells_rank * bytes_per_cell / (kernel_ms_step * 1e-3) / 1e9) if kernel_ms_step > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (cells_rank * bytes_per_cell / (kernel_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms_step > 0 else None,
                         "traffic": None, "bytes_per_cell": bytes_per_cell,
                         "note": f"rank 0's {pure.launches_per_step} launch(es) per pure-sharding step ({n_rank_refs} reference views), HIP start/stop events of each launch"},
        }
        if sampled is not None:
            line["sampled_mode"] = sampled
        if world > 1:
            line["scaling_note"] = ("no multi-GPU node was available to the builder: when this line comes from N real GPUs it is the first measurement. In dense mode ONE "
                                    "GPU triangulates the scene faster than its survivors can cross an xGMI link (DESIGN 5): `value_pure_sharding` is bound by the link, "
                                    "`value` recomputes what is cheaper to recompute than to move (the plan's inputs are measured in this run); `sampled_mode` is the "
                                    "upstream-equivalent job, whose exchange is small")
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    dens.close()


def end_to_end(args):
    """SURVEY 8d iii: points/s INCLUDING the RoMa-v2 forward, on a real scene.  Runs only where everything it needs exists; otherwise (None, why)."""
    missing = []
    if not args.scene_root:
        missing.append("no --scene-root / LFD_SCENE_ROOT (a COLMAP scene with images)")
    elif not os.path.isdir(os.path.join(args.scene_root, "sparse", "0")):
        missing.append(f"{args.scene_root}/sparse/0 not found")
    try:
        from lichtfeld_densification_plugin_amd.core import matcher as mm
        mm._import_romav2()
    except Exception as exc:
        missing.append(f"RoMa-v2 is not importable ({str(exc).splitlines()[0][:120]})")
        mm = None
    weights = os.environ.get("LFD_ROMA_WEIGHTS")
    if mm is not None and not (weights and os.path.isfile(weights)) and not mm.has_cached_romav2_weights():
        missing.append("romav2.pt not found (torch hub cache / LFD_ROMA_WEIGHTS)")
    if missing:
        return None, "points/s including the RoMa-v2 forward (SURVEY 8d iii) is unmeasured: " + "; ".join(missing)
    from lichtfeld_densification_plugin_amd import densify
    if weights:                                    # where torch.hub looks for it
        import shutil
        dst = os.path.join(torch.hub.get_dir(), "checkpoints", "romav2.pt")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if not os.path.exists(dst):
            shutil.copyfile(weights, dst)
    argv = ["--scene_root", args.scene_root, "--roma_setting", args.preset, "--out_name", "points3D_dense_bench.ply"]
    for sub in ("images_4", "images_2", "images"):
        if os.path.isdir(os.path.join(args.scene_root, sub)):
            argv += ["--images_subdir", sub]
            break
    a = densify.build_argparser().parse_args(argv)
    seen = {}
    t0 = time.perf_counter()
    rc = densify.dense_init(a, progress_callback=lambda p, m: seen.__setitem__("last", m))
    dt = time.perf_counter() - t0
    if rc != 0:
        return None, f"dense_init returned {rc}"
    n = int(str(seen.get("last", "0")).split("!")[-1].split("points")[0].replace(",", "").strip() or 0)
    return {"points_per_s": n / dt, "seconds": dt, "points": n, "scene": args.scene_root, "setting": args.preset,
            "note": "densify.dense_init (CLI defaults): COLMAP read, image loading, RoMa-v2 forward, sampled-mode hot path, PLY written"}, None


def dry_run(args):
    """`--gpus N --dry-run`: the sharded benchmark's plan, rank by rank, from the arguments alone (no GPU, no communicator).  The collective
    sequence comes from core/distributed.py::exchange_schedule, which tests/test_distributed_cpu.py compares operation by operation with the
    calls a gloo run of the same exchange really makes: the first run on real hardware either issues exactly this list on every rank or the
    diff says where it departed.  Sizes use the workload's nominal survivor fraction (0.87 of the cells)."""
    from lichtfeld_densification_plugin_amd.core import distributed as lfd_dist
    world = int(args.gpus)
    h_lr, w_lr, H, W = synthetic.ROMA_PRESETS[args.preset]
    total_refs = args.refs * world if args.scaling == "weak" else args.refs
    ply = args.exchange_records == "ply"
    cols, rec_bytes = (15, 15) if ply else (7, 28)
    n_rep = 0 if args.replicate == "auto" else max(0, min(int(args.replicate), total_refs))
    s_nominal = 0.87
    per_ref = int(H * W * s_nominal)
    ranks = []
    per_round = n_rounds = n_sh = 0
    for rank in range(world):
        consumes = args.exchange == "all_gather" or rank == 0
        mine, n_sh = lfd_dist.split_replicated(total_refs, n_rep, rank, world, replicas_here=consumes)
        sh = [g for g in mine if g < n_sh]
        rp = [g for g in mine if g >= n_sh]
        n_local_max = (n_sh + world - 1) // world
        per_round = max(1, -(-n_local_max // max(1, int(args.exchange_rounds)))) if n_sh else 1
        n_rounds = -(-n_local_max // per_round) if n_sh else 0
        chunks = [sh[c * per_round:(c + 1) * per_round] for c in range(n_rounds)]
        ranks.append({"rank": rank, "device": f"cuda:{rank}", "sharded_positions": sh, "replicated_positions": rp, "consumes_cloud": consumes,
                      "launches_per_step": sum(1 for ch in chunks if ch) + (1 if rp else 0),
                      "launches": [{"round": c, "references": ch, "record_buffer_bytes": len(ch) * H * W * rec_bytes,
                                    "inputs_bytes": len(ch) * (args.k * H * W * 12 + h_lr * w_lr * 3)} for c, ch in enumerate(chunks)],
                      "cloud_buffer_bytes": ((n_sh + len(rp)) * H * W * rec_bytes) if consumes else 0,
                      "nominal_records_sent_per_step": len(sh) * per_ref * rec_bytes})
    plan = lfd_dist.exchange_schedule(n_sh, world, per_round, args.exchange, "ply" if ply else "f32", counts=[per_ref] * n_sh, eager=True)
    out = {"dry_run": True, "n_gpus": world, "workload": WORKLOADS[args.workload]["what"], "scaling": args.scaling, "refs_total": total_refs, "neighbours": args.k,
           "grid": [H, W], "exchange": {"form": args.exchange, "record_bytes": rec_bytes, "rounds": plan["rounds"], "refs_per_round": plan["refs_per_round"],
                                        "eager": True, "replicated": n_rep,
                                        "replicated_note": "`--replicate auto` measures its candidates at run time: the plan shows pure sharding" if args.replicate == "auto" else None},
           "ranks": ranks,
           "per_step_collectives": plan["collectives"],
           "around_the_timed_region": ["barrier (torch.cuda.synchronize + dist.barrier) before and after the K steps", "all_reduce(MAX) of the elapsed times",
                                       "all_reduce(SUM) of the survivor counts"],
           "backend": "nccl (RCCL over xGMI), one process per GPU, rendezvous 127.0.0.1",
           "note": "every rank issues per_step_collectives in exactly this order, K + warm-up times; numel_* are per rank, in elements of the tensor handed over"}
    print(json.dumps(out), flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes - this parent has not touched
    the GPU and never will - wait for them, pass rank 0's JSON line through."""
    n_dev = torch.cuda.device_count()          # does not initialise the GPU runtime on this image
    share = int(os.environ.get("LFD_BENCH_RANKS_PER_GPU", "1"))
    if n_dev * share < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) visible"
                         + (" (LFD_BENCH_RANKS_PER_GPU lets several ranks share one for a functional check)" if share == 1 else ""))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import tempfile
    procs, logs = [], []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # every rank's output goes to files of its own (a pipe nobody drains would block a chatty rank); only fresh children are started, this
        # process never becomes one of them
        logs.append((tempfile.TemporaryFile(mode="w+"), tempfile.TemporaryFile(mode="w+")))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=logs[-1][0], stderr=logs[-1][1], text=True))
    # ALL children are watched: when one exits non-zero the others - who would sit in a collective until the backend's timeout - are terminated
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)

    def tail(fh, n=2000):
        fh.seek(0)
        return fh.read()[-n:]
    logs[0][0].seek(0)
    sys.stdout.write(logs[0][0].read())
    sys.stdout.flush()
    if any(codes):
        for r, (_o, e) in enumerate(logs):
            t = tail(e).strip()
            if t:
                sys.stderr.write(f"---- rank {r} (exit {codes[r]}) stderr tail ----\n{t}\n")
        raise SystemExit(f"rank exit codes {codes}")


def main():
    args = parse_args()
    if args.dry_run:
        return dry_run(args)                      # nothing below is reached: no GPU, no communicator
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)                 # before anything here touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (or run `python bench.py --gpus {args.gpus}` "
                         "without a launcher, which starts them itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # torch sizes its intra-op pool by the CPUs it sees (256 on the pool's boxes), the container may use 16: the spinning workers would burn the CFS
    # quota and the kernel would stop the whole process for most of every 100 ms (profiles/r4/sampled_sustained.txt)
    from lichtfeld_densification_plugin_amd.core import hostenv
    host_threads = hostenv.fit_threads_to_quota()
    args.host = {"torch_threads": host_threads, "cpu_quota_cores": hostenv.cpu_quota(), "cpus_visible": os.cpu_count()}
    share = max(1, int(os.environ.get("LFD_BENCH_RANKS_PER_GPU", "1")))
    shared_gpu = share > 1                        # functional check: several ranks on one GPU -> gloo (own launcher or torch.distributed.run alike)
    local_rank //= share
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1 or os.environ.get("LFD_BENCH_FORCE_DIST"):     # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = "gloo" if shared_gpu else "nccl"
        # (a collective that never completes - the first run on real multi-GPU hardware is still ahead - becomes an error after five minutes, not a
        # silent wait until the backend's own half-hour default)
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("LFD_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)

    if dist is not None:
        return run_sharded(args, world, rank, dev, dist, backend)
    cams, refs, srefs, dims, mine, total_refs = build_workload(args, rank, world, dev)
    H, W, wm, hm = dims
    cfg = lfd.DensePipelineConfig(output_path="", roma_setting=args.preset, nns_per_ref=args.k)   # GUI defaults
    params = hb.make_params(cfg)
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    if total_refs < world:        # (the same verdict on every rank, before any collective: nobody is left waiting in a barrier)
        raise SystemExit(f"--refs {args.refs} ({args.scaling} scaling) leaves a rank without a reference view: {total_refs} references over {world} ranks")
    # DISTINCT batches rotate in the timed loop (the same planes, the references in another order: other descriptor tables,
    # other per-pair constants), as in a real run where every launch sees a new batch: the descriptor upload and
    # lfd_pair_setup_kernel are then inside the timed region.  They are issued by lfd_prepare_batch - which stages them on the
    # context's preparation stream, beside the kernels of the batch before - so that the HIP events around the launch that
    # follows bracket the dense kernel alone.
    batch = hb.PreparedBatch(refs, wm, hm, cameras=cams)      # the pipeline's default: upstream's own F handed to the kernels
    # (three: the context keeps the tables of the last TWO batches on the device, so that the next one can be staged while the current
    # one computes; with fewer than three distinct batches in rotation nothing would ever be uploaded again)
    n_rot = 1 if (args.cached_batch or len(refs) < 3) else 3
    batches = [batch] + [hb.PreparedBatch(refs[j:] + refs[:j], wm, hm, cameras=cams) for j in range(1, n_rot)]
    out = hb.OutputBuffers(len(refs) * H * W, len(refs), args.k, dev, with_cell=False, with_segments=False)   # upstream emits xyz, rgb, err only

    def barrier():
        torch.cuda.synchronize(dev)

    def step(i, evs=None):
        b = batches[i % len(batches)]
        if evs is not None:
            evs[0].record()
        dens.prepare(b, params)         # tables + per-pair constants of THIS batch (a no-op when it is the batch of the last launch)
        if evs is not None:
            evs[1].record()             # same stream the kernel is launched on (torch's current stream)
        dens.launch_dense(b, params, out)
        if evs is not None:
            evs[2].record()

    # untimed spin-up (clocks, first-touch of the output pages), then the W warm-up steps
    t_spin = time.perf_counter()
    n_spin = 0
    while time.perf_counter() - t_spin < args.spinup_s:       # back-to-back like the timed region, so that the power management has settled
        for _ in range(32):
            step(n_spin); n_spin += 1
        torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    dens.time_dense_kernels(args.steps)       # (creates the timed launches' event pairs: before the barrier, not inside the timed region)
    barrier()
    # The timed region: EXACTLY K steps, nothing else in the stream.  Every launch carries a start and a stop event of its own
    # (lfd_kernel_timing: hipExtLaunchKernelGGL, recorded on the launch stream where the kernel begins and ends - what a kernel
    # trace reports): the dense kernel's duration is measured live, inside the timed region, without a packet that is not the step's.
    t0 = time.perf_counter()
    host_t = []
    for i in range(args.steps):
        step(i)
        host_t.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("LFD_BENCH_DEBUG"):
        print(f"[rank {rank}] enqueue done at {t_enq * 1e3:.2f} ms, first 5 host stamps {[round(x * 1e3, 2) for x in host_t[:5]]}", file=sys.stderr)
    per_launch = [float(x) for x in dens.dense_kernel_times_ms()]
    dens.time_dense_kernels(0)
    if len(per_launch) != args.steps:
        raise SystemExit(f"{len(per_launch)} kernel timings for {args.steps} steps")
    kernel_ms = float(np.mean(per_launch))
    # A second pass of the same K steps, outside the value: events recorded AROUND the two calls of a step - what is left of the batch
    # preparation in the launch stream (`fresh_batch_ms`) and the launch as a caller's events see it (`kernel_ms_bracketed`: the
    # kernel plus the dispatch gap in front of it).  Three more packets per step, which is why this is not the timed region.
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    for i, e3 in enumerate(ev):
        step(i, e3)
    barrier()
    per_bracket = [e1.elapsed_time(e2) for _e0, e1, e2 in ev]
    per_prepare = [e0.elapsed_time(e1) for e0, e1, _e2 in ev]
    kernel_ms_bracketed = float(np.mean(per_bracket))
    fresh_batch_ms = float(np.mean(per_prepare))
    kernel_pct = {"p50": float(np.percentile(per_launch, 50)), "p95": float(np.percentile(per_launch, 95)),
                  "min": float(np.min(per_launch)), "max": float(np.max(per_launch))}
    if os.environ.get("LFD_BENCH_DEBUG"):
        print(f"[rank {rank}] per-launch us: " + " ".join(f"{x * 1e3:.0f}" for x in per_launch), file=sys.stderr)
        print(f"[rank {rank}] per-launch ms: min {min(per_launch):.3f} max {max(per_launch):.3f} mean {kernel_ms:.3f}; "
              f"wall {elapsed * 1e3:.2f} ms for {args.steps} steps", file=sys.stderr)
    dens.check_launches()
    res = out.collect()
    n_pts = res.count
    rot = (args.steps - 1) % len(batches) if args.steps > 0 else 0      # the last launch's references are `refs` rotated by this much

    total_pts = float(n_pts)

    if rank == 0:
        cells = len(refs) * H * W
        s_frac = n_pts / cells
        bytes_per_cell = 4 * args.k + 11 + 28 * s_frac
        algo_bytes = cells * bytes_per_cell
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        value = total_pts * args.steps / elapsed
        traffic, traffic_source = traffic_bytes(args)
        rccl = 0                                   # ranks the collective backend reports: no communicator exists on the N = 1 line
        line = {
            "metric": "triangulated points/sec + pairs/sec, MipNeRF360 garden @fast, 1/2/4/8 GPU",     # BASELINE.json's metric; value = points/s, pairs_per_s beside it
            "value": value, "unit": "points/s", "n_gpus": world, "rccl_ranks": rccl, "collective_backend": backend, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32 (+f64 Sampson/DLT solve)", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.workload]['what']}: ring of {WORKLOADS[args.workload]['n_cams']} cameras {WORKLOADS[args.workload].get('width', 1297)}x{WORKLOADS[args.workload].get('height', 840)}, `{args.preset}` grid {H}x{W}, "
                                   f"{args.refs} reference views x {args.k} neighbours {'resident per GPU' if args.scaling == 'weak' else 'in total, dealt over the ranks'}, "
                                   f"default thresholds (certainty 0.2 / sampson 5.0 / reproj 0.8 / parallax 0.5 deg), "
                                   f"noise {args.noise_px} px, {args.outliers:.0%} outliers",
                       "kernel": "fused dense filter+triangulate kernel (lfd_dense_kernel)", "mode": "dense", "refs_per_gpu": len(refs), "refs_total": total_refs,
                       "neighbours": args.k, "grid": [H, W], "sharding": f"references round-robin over {world} rank(s)",
                       "batches_in_rotation": len(batches)},
            "pairs_per_s": total_refs * args.k * args.steps / elapsed,
            "cells_per_s": total_refs * H * W * args.steps / elapsed,
            "survivor_fraction": s_frac,
            # one step = lfd_prepare_batch (descriptor upload + per-pair constants of a batch the context has not just seen) + the dense kernel
            "compute_ms": elapsed / args.steps * 1e3, "kernel_ms": kernel_ms, "fresh_batch_ms": fresh_batch_ms,
            "host": getattr(args, "host", None),
            "end_to_end": None,
            "end_to_end_note": "points/s including the RoMa-v2 forward (SURVEY 8d iii) is unmeasured: neither the RoMa-v2 weights nor torchvision are on the box",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "lfd_dense_kernel", "kernel_ms": kernel_ms, "kernel_ms_percentiles": kernel_pct,
                         "kernel_ms_source": "HIP start/stop events of each timed launch (hipExtLaunchKernelGGL) on the launch stream, mean over the timed region",
                         "kernel_ms_bracketed": kernel_ms_bracketed,      # events recorded around the call instead: they include the dispatch gap before the kernel
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_cell": bytes_per_cell},
        }
        # the same time priced against the byte count of a kernel that reads the warps of ALL k neighbours coalesced
        # (12k + 3 + 28 s per cell) - what lfd_dense_kernel physically does for references with at most LFD_DENSE_ALL_WARPS (2) neighbours
        line["roofline"]["achieved_all_warps_bytes"] = cells * (12 * args.k + 3 + 28 * s_frac) / (kernel_ms * 1e-3) / 1e9
        if not args.light and world == 1:
            line["roofline"]["device_copy_GBps"] = device_copy_bandwidth(dev)      # what a plain device-to-device copy reaches on this box
        line["roofline"]["valu_busy_frac"] = valu_busy_frac(args)     # the f64 geometry makes the kernel vector-ALU-bound, not HBM-bound (same source as `traffic`)
        line["roofline"]["valu"] = valu_roofline(args, kernel_ms)      # ... and that bound itself: issue time of the vector instructions / kernel time
        line["roofline"]["traffic_over_algorithmic"] = (line["roofline"]["traffic"] / algo_bytes) if line["roofline"]["traffic"] else None
        if not args.light and world == 1:
            e2e, why = end_to_end(args)
            line["end_to_end"] = e2e
            if why:
                line["end_to_end_note"] = why
            else:
                line.pop("end_to_end_note", None)
            line["roofline"]["power"] = power_under_kernel(dens, batches, params, out, dev)      # the bound that binds: the package's power cap
            line["unordered_retirement"] = unordered_rate(args, dens, batch, params, out, H, W, algo_bytes)
            line["ply_output"] = ply_output_rate(args, dens, batch, params, H, W, cells, s_frac)
        if not args.light and world == 1:     # the side legs (and the CPU baseline) belong to the N = 1 line; at N > 1 the other ranks are waiting
            line["with_d2h"] = d2h_inclusive_rate(dens, batch, params, out, dev)
            line["sampled_mode"] = sampled_mode_rate(args, dens, refs, dims, cfg, cams_for_hot=cams)
            line["secondary_kernels"] = secondary_kernels(args, dens, batch, refs, dims, cfg, res)
            par = parity_report(args, dens, cams, refs, srefs, dims, cfg)
            if par is not None:
                line["parity"] = par
            base = cpu_baseline(args, cams, srefs, dims, cfg)
            if base is not None:
                line["cpu_baseline"] = base
            # the user-visible default beside the dense headline: triangulation_mode="sampled", upstream's normaliser, refs_per_launch = 0 = automatic -
            # sixteen references per fused call on upstream's ONE MT19937 stream (lfd_triangulate_sampled_chain: the cells, points and stream of
            # successive single calls, bit for bit) when nobody waits for intermediate previews (the CLI), one reference per call when somebody
            # does (the GUI's previews, debug states): both are reported
            sm = line["sampled_mode"]
            dflt_ms = sm.get("default_config_ms_per_reference")
            ch = sm.get("chained") or {}
            if dflt_ms and ch.get("default_config_ms_per_reference"):
                cms = ch["default_config_ms_per_reference"]
                line["default_mode"] = {"mode": f"sampled (upstream-equivalent), upstream_normaliser, refs_per_launch=0 (automatic: {ch['references_per_call']} references per fused call on "
                                                "the one stream, lfd_triangulate_sampled_chain)",
                                        "ms_per_reference": cms, "refs_per_s": 1e3 / cms, "pairs_per_s": 1e3 / cms * args.k,
                                        "points_per_s": ch["default_config_points_per_reference"] * 1e3 / cms,
                                        "ms_per_reference_device_sums": ch["ms_per_reference"],
                                        "one_reference_per_call": {"mode": "the same with refs_per_launch=1 (what a run with intermediate previews uses)", "ms_per_reference": dflt_ms,
                                                                   "refs_per_s": 1e3 / dflt_ms, "pairs_per_s": 1e3 / dflt_ms * args.k,
                                                                   "points_per_s": sm["default_config_points_per_reference"] * 1e3 / dflt_ms}}
            if args.pipeline_cams > 0:
                import bench_pipeline
                dens.close()                        # the leg builds its own contexts
                pw, ph = (int(v) for v in args.pipeline_size.split("x"))
                line["pipeline"] = bench_pipeline.pipeline_leg(dev, n_cams=args.pipeline_cams, latency_ms=args.pipeline_latency_ms, roma_setting=args.preset,
                                                               width=pw, height=ph)
        print(json.dumps(line), flush=True)
    dens.close()


if __name__ == "__main__":
    main()
