#!/usr/bin/env python3
"""Predicted 1 -> 8 GPU curve of BASELINE config 4 (one scene: 56 references x 8 neighbours at `fast`), dense and sampled mode.
NOT a measurement of N GPUs (a gpurun box has one): per-rank compute is MEASURED on one MI355X at the rank's share of the work
(profiles/run_scaling_inputs.sh), the exchange is MODELLED with SURVEY 8e's link model:

    xGMI: 7 links per GPU, one per peer, ~153 GB/s per direction each (taken at 80 % = 122 GB/s achievable), ~25 us per collective;
    all-gather, direct (every rank sends its shard to every peer on that peer's own link): shard_bytes / link_bw;
    all-gather, ring (what a bandwidth-optimal ring costs when one link bounds each step): (N - 1) * shard_bytes / link_bw;
    gather to rank 0: the root receives N - 1 shards on N - 1 links at once: shard_bytes / link_bw.
Usage: python profiles/scaling_model.py gpurun_out/scaling_inputs.jsonl"""
import json
import sys

LINK = 153e9 * 0.8
LAT = 25e-6
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for mode in ("dense", "sampled"):
    sel = sorted([r for r in rows if r["mode"] == mode], key=lambda r: r["ranks"])
    if not sel:
        continue
    t1 = sel[0]["step_ms"]
    total_pts = sel[0]["points"]
    print(f"\n{mode} mode, config 4 (56 references x 8 neighbours, fast): {total_pts / 1e6:.2f} M surviving points = {total_pts * 28 / 1e6:.1f} MB to exchange")
    print("ranks  refs/rank  compute ms (measured, 1 GPU)   exchange ms (model: direct | ring | to root)   step+exchange ms (direct)   speed-up vs 1 GPU   points/s incl. exchange")
    for r in sel:
        n = r["ranks"]
        shard = total_pts * 28 / n
        if n == 1:
            direct = ring = root = 0.0
        else:
            direct = (2 * LAT + shard / LINK) * 1e3
            ring = (2 * LAT + (n - 1) * shard / LINK) * 1e3
            root = (2 * LAT + shard / LINK) * 1e3
        tot = r["step_ms"] + direct
        print(f"{n:5d}  {r['refs_per_rank']:9d}  {r['step_ms']:28.3f}   {direct:10.3f} | {ring:7.3f} | {root:7.3f}                 {tot:12.3f}   {t1 / tot:17.2f}   {total_pts / (tot * 1e-3):.3e}")
