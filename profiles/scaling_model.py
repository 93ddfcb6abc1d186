#!/usr/bin/env python3
"""Predicted 1 -> 8 GPU curve of BASELINE config 4 (one scene: 56 references x 8 neighbours at `fast`), dense and sampled mode.
NOT a measurement of N GPUs (a gpurun box has one): per-rank compute is MEASURED on one MI355X at the rank's share of the work
(profiles/run_scaling_inputs.sh), the exchange is MODELLED with SURVEY 8e's link model:

    xGMI: 7 links per GPU, one per peer, ~153 GB/s per direction each (taken at 80 % = 122 GB/s achievable), ~25 us per collective;
    all-gather, direct (every rank sends its shard to every peer on that peer's own link): shard_bytes / link_bw;
    all-gather, ring (what a bandwidth-optimal ring costs when one link bounds each step): (N - 1) * shard_bytes / link_bw;
    gather to rank 0: the root receives N - 1 shards on N - 1 links at once: shard_bytes / link_bw.
Round 4 adds the OVERLAPPED schedule of core/distributed.py::OverlappedExchange / bench.py --gpus N: the rank's share in R rounds, round c's
records (28-byte rows or 15-byte PLY vertex records) in an asynchronous collective beside the compute of round c + 1, the last round's
exchange exposed:  t = c + (R - 1) max(c, x) + x  with c = compute / R and x = 2 LAT + shard_bytes / R / link_bw.
Usage: python profiles/scaling_model.py gpurun_out/scaling_inputs.jsonl [rounds]"""
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


ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print(f"\n==== overlapped schedule, {ROUNDS} rounds per rank (direct all-gather / gather to root: one shard per link either way) ====")
for mode in ("dense", "sampled"):
    sel = sorted([r for r in rows if r["mode"] == mode], key=lambda r: r["ranks"])
    if not sel:
        continue
    t1 = sel[0]["step_ms"]
    total_pts = sel[0]["points"]
    print(f"\n{mode} mode: {total_pts / 1e6:.2f} M surviving points = {total_pts * 28 / 1e6:.1f} MB as 28-byte rows, {total_pts * 15 / 1e6:.1f} MB as 15-byte PLY records")
    print("ranks  compute ms   28 B end of run (round 3)   28 B overlapped   15 B overlapped   speed-up (28 B eor | 28 B ovl | 15 B ovl)   bound at 15 B   counts only: ms, speed-up (cloud left sharded in HBM)")
    for r in sel:
        n = r["ranks"]
        comp = r["step_ms"]
        res = []
        for nbytes, overlapped in ((28, False), (28, True), (15, True)):
            if n == 1:
                res.append(comp)
                continue
            shard = total_pts * nbytes / n
            if not overlapped:
                res.append(comp + (2 * LAT + shard / LINK) * 1e3)
            else:
                c = comp / ROUNDS
                x = (2 * LAT + shard / ROUNDS / LINK) * 1e3
                res.append(c + (ROUNDS - 1) * max(c, x) + x)
        link_ms = 0.0 if n == 1 else total_pts * 15 / n / LINK * 1e3
        bound = "compute" if n == 1 or comp >= link_ms else f"link ({link_ms:.3f} ms to move one shard)"
        resident = comp if n == 1 else comp + LAT * 1e3          # the rounds' count all-gathers ride beside the launches; the last one is waited for
        print(f"{n:5d}  {comp:10.3f}   {res[0]:25.3f}   {res[1]:15.3f}   {res[2]:15.3f}   {t1 / res[0]:10.2f} | {t1 / res[1]:8.2f} | {t1 / res[2]:8.2f}              {bound:42s} {resident:.3f}  {t1 / resident:.2f}")


# ---- round 4: recompute instead of communicate (core/distributed.py::plan_replication) ---------------------------------------------------------------
# the same inputs through the planner bench.py uses: per-reference compute and launch cost from the measured dense rows (a line through the
# 56 / 28 / 14 / 7-reference launches), 15-byte records, one link per peer, 0.1 ms of collectives per round trip, the placement copy at 3.5 TB/s
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
try:
    from lichtfeld_densification_plugin_amd.core import distributed as _d
    dense = sorted([r for r in rows if r["mode"] == "dense"], key=lambda r: r["ranks"])
    if len(dense) >= 2:
        n_total = dense[0]["refs_per_rank"] if "refs_per_rank" in dense[0] else 56
        xs = [r.get("refs_per_rank", n_total // r["ranks"]) for r in dense]
        ys = [r["step_ms"] for r in dense]
        mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
        b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        a = my - b * mx
        ref_bytes = dense[0]["points"] * 15 / n_total
        print(f"\n==== replicated suffix + sharded prefix (plan_replication): launch {a:.4f} ms + {b:.5f} ms per reference, {ref_bytes / 1e6:.2f} MB of records per reference ====")
        print("ranks  sharded  replicated   step ms   speed-up vs 1 GPU   | pure sharding: ms, speed-up")
        for n in (1, 2, 4, 8):
            p = _d.plan_replication(n_total, n, b, ref_bytes, launch_ms=a, link_gbps=LINK / 1e9, collective_ms=0.1, copy_gbps=3500.0)
            print(f"{n:5d}  {p['n_sharded']:7d}  {p['n_replicated']:10d}   {p['step_ms']:.3f}     {p['single_rank_ms'] / p['step_ms']:.2f}                | "
                  f"{p['pure_sharding_ms']:.3f}  {p['single_rank_ms'] / p['pure_sharding_ms']:.2f}")
except ImportError as exc:          # (the model's first part needs nothing but the input file)
    print(f"(replication plan skipped: {exc})")
