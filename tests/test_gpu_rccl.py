"""The exchange helpers of core/distributed.py through RCCL itself (backend "nccl", device buffers) - one rank, which is what one
GPU allows: the communicator is created, the counts and the records travel through RCCL's all-gather on device memory, the
point-to-point paths see the "nccl" branch (device-resident buffers, no host staging).  The N > 1 behaviour of the same functions is
covered on CPU (gloo, 2-4 ranks: tests/test_distributed_cpu.py); the 8-GPU run is the driver's."""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import importlib, io, os, sys
    import numpy as np, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    lfd = importlib.import_module("lichtfeld-densification-plugin_amd")
    D = importlib.import_module("lichtfeld-densification-plugin_amd.core.distributed")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1, device_id=dev)
    try:
        assert D._backend_name(dist) == "nccl"
        g = torch.Generator().manual_seed(5)
        counts = [7, 0, 1200, 33]
        n = sum(counts)
        xyz = torch.randn(n, 3, generator=g).to(dev); rgb = torch.rand(n, 3, generator=g).to(dev); err = torch.rand(n, generator=g).to(dev)
        assert D._collective_device(xyz, dist) == dev                       # RCCL: the records never leave the device
        x, c, e, gc = D.all_gather_by_reference(xyz, rgb, err, counts, len(counts), dist)
        assert x.device == dev and torch.equal(x, xyz) and torch.equal(c, rgb) and torch.equal(e, err) and gc.tolist() == counts
        x, c, e, gc = D.gather_to_root_by_reference(xyz, rgb, err, counts, len(counts), dist)
        assert x.device == dev and torch.equal(x, xyz) and torch.equal(c, rgb) and torch.equal(e, err) and gc.tolist() == counts
        x, c, e, cn = D.all_gather_points(xyz, rgb, err, dist)
        assert torch.equal(x, xyz) and cn == [n]
        assert D.agree_on_status(3, dist, device=dev) == 3
        e0 = torch.zeros(0, 3, device=dev)
        x, c, e, gc = D.all_gather_by_reference(e0, e0, torch.zeros(0, device=dev), [0, 0], 2, dist)      # nothing survived anywhere
        assert x.shape == (0, 3) and gc.tolist() == [0, 0]

        class Sink:                      # a StreamedPlyWriter stand-in: collects what rank 0 would append to the file
            def __init__(self): self.parts = []
            def append_packed(self, b): self.parts.append(bytes(b))
        sink = Sink()
        stream = D.ShardedPlyStream(dist, 3, sink, dev)
        assert stream.dev == dev
        bodies = [torch.arange(15 * k, dtype=torch.uint8, device=dev) for k in (4, 0, 9)]
        for i, b in enumerate(bodies):
            if b.numel(): stream.push(i, b)
        stream.finish()
        assert b"".join(sink.parts) == b"".join(b.cpu().numpy().tobytes() for b in bodies)
        # round 5 (core/sinks.py::ShardLink): the streamed file travels on a process group OF ITS OWN, which is destroyed at the end of the run, and the
        # result is exchanged on the default group only AFTER that stream has been drained - never two communicators in flight at once
        grp = dist.new_group()
        sink2 = Sink()
        stream2 = D.ShardedPlyStream(dist, 3, sink2, dev, group=grp)
        for i, b in enumerate(bodies):
            if b.numel(): stream2.push(i, b)
        stream2.finish()
        assert b"".join(sink2.parts) == b"".join(sink.parts)
        dist.destroy_process_group(grp)
        x, c, e, gc = D.all_gather_by_reference(xyz, rgb, err, counts, len(counts), dist)          # the default communicator is untouched by that
        assert torch.equal(x, xyz) and gc.tolist() == counts
        plan = D.exchange_schedule(len(counts), 1, 2, "all_gather", "f32", counts=counts)
        rec_dist = D.RecordingDist(dist)
        ex = D.OverlappedExchange(rec_dist, len(counts), 2, dev, form="all_gather", record="f32")
        for i, c_ in enumerate(counts):
            if c_: ex.push(i, torch.zeros(c_, 7, device=dev))
        ex.finish()
        issued = [e_ for e_ in rec_dist.log if e_["op"] in ("all_gather_into_tensor", "gather")]
        assert [(e_["op"], e_["numel_in"]) for e_ in issued] == [(p_["op"], p_["numel_in"]) for p_ in plan["collectives"]], (issued, plan["collectives"])
        # the exchange in rounds (asynchronous collectives on the communicator's stream), both forms and both record kinds
        parts = [torch.randn(c, 7, generator=g).to(dev) for c in counts]
        for form in ("all_gather", "gather_to_root"):
            ex = D.OverlappedExchange(dist, len(counts), 2, dev, form=form, record="f32")
            assert ex.dev == dev and ex.n_rounds == 2
            for i, p_ in enumerate(parts):
                if p_.shape[0]: ex.push(i, p_)
            recs, gc = ex.finish()
            assert recs.device == dev and torch.equal(recs, torch.cat(parts)) and gc.tolist() == counts
        recs15 = [torch.randint(0, 255, (c * 15,), generator=g, dtype=torch.uint8).to(dev) for c in counts]
        ex = D.OverlappedExchange(dist, len(counts), 1, dev, form="all_gather", record="ply")
        for i, p_ in enumerate(recs15):
            if p_.numel(): ex.push(i, p_)
        recs, gc = ex.finish()
        assert torch.equal(recs, torch.cat(recs15)) and gc.tolist() == counts and ex.n_rounds == 4
        print("RCCL_OK", torch.cuda.get_device_name(0))
    finally:
        dist.destroy_process_group()
""") % REPO


@pytest.mark.gpu
def test_exchange_helpers_on_device_buffers_through_rccl(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-c", SCRIPT, str(port)], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.gpu
def test_bench_exchange_legs_through_rccl_on_one_rank():
    """bench.py with LFD_BENCH_FORCE_DIST=1 creates the RCCL communicator for its one rank and runs the sharded leg on device buffers through
    it - the rounds of the overlapped exchange (counts + padded records, asynchronous collectives on the communicator's stream) and the
    end-of-run exchanges: the line then says `collective_backend: nccl`."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LFD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (the third run: two of the six references "replicated" - their records written in place behind the exchanged part of the one cloud buffer,
    # the exchange placing its ordered records in front of them: the placement path device collectives take, bench.py asserts the cloud's bytes)
    for extra in ([], ["--exchange", "gather_to_root", "--exchange-records", "f32", "--exchange-rounds", "3"], ["--replicate", "2"]):
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "config4", "--refs", "6",
               "--preset", "turbo", "--light", "--spinup-s", "0.05"] + extra
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
        assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
        d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl" and d["rccl_ranks"] == 1       # (what the communicator itself reports through an all-reduce)
        ex = d["exchange"]
        assert ex["points"] > 0 and ex["overlapped"] and ex["end_of_run_28B"]["allgather_ms"] > 0 and ex["end_of_run_28B"]["gather_to_root_ms"] > 0
        assert ex["rounds"] == (3 if "gather_to_root" in extra else 2) and ex["record_bytes"] == (28 if "f32" in extra else 15)
        assert 0 < d["value"] and d["value_compute_only"] > 0 and d["value_sharded_resident"] > 0
        if "--replicate" in extra:
            assert d["replication"]["n_replicated"] == 2 and d["replication"]["forced"] and d["value_pure_sharding"] > 0
        else:
            assert d["replication"]["n_replicated"] == 0 and d["value"] == d["value_pure_sharding"]        # one rank: nothing to exchange, nothing to replicate
