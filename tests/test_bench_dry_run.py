"""`bench.py --gpus N --dry-run`: the sharded benchmark's plan from the arguments alone - no GPU (this test runs where there is none), no communicator.
Its collective list is core/distributed.py::exchange_schedule, which tests/test_distributed_cpu.py::test_overlapped_exchange_is_the_single_process_sequence
compares call by call with what a gloo run of the same exchange issues; here: the plan is complete and self-consistent for the driver's launch forms
(N = 2, 4, 8; SURVEY 8e: references round-robin over the ranks, one exchange of the survivors, every rank the same sequence)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                     # even on a GPU box the dry run must not need one
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", *argv], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4, 8])
def test_the_default_sharded_schedule(n):
    d = _plan("--gpus", str(n))
    assert d["dry_run"] and d["n_gpus"] == n and d["refs_total"] == 56 and d["neighbours"] == 8 and d["scaling"] == "strong"      # BASELINE config[3]
    seen = []
    for r, rk in enumerate(d["ranks"]):
        assert rk["rank"] == r and rk["sharded_positions"] == list(range(r, 56, n)) and rk["replicated_positions"] == []
        assert [g for la in rk["launches"] for g in la["references"]] == rk["sharded_positions"]
        assert all(la["record_buffer_bytes"] == len(la["references"]) * 512 * 512 * 15 for la in rk["launches"])
        assert rk["cloud_buffer_bytes"] == 56 * 512 * 512 * 15 and rk["launches_per_step"] == len([la for la in rk["launches"] if la["references"]])
        seen += rk["sharded_positions"]
    assert sorted(seen) == list(range(56))                       # every reference on exactly one rank
    ex, seq = d["exchange"], d["per_step_collectives"]
    assert ex["rounds"] == 2 and ex["record_bytes"] == 15 and ex["eager"]
    # eager rounds: counts, then records, round after round; the same list for every rank (it is printed once)
    assert [(c["what"], c["round"]) for c in seq] == [("counts", 0), ("records", 0), ("counts", 1), ("records", 1)]
    for c in seq:
        assert c["op"] == "all_gather_into_tensor" and c["numel_out"] == n * c["numel_in"]
    assert seq[0]["numel_in"] == ex["refs_per_round"] and seq[1]["dtype"] == "uint8" and seq[1]["numel_in"] == seq[1]["rows"] * 15


def test_gather_to_root_replication_and_weak_scaling():
    d = _plan("--gpus", "4", "--exchange", "gather_to_root", "--exchange-records", "f32", "--replicate", "8", "--exchange-rounds", "3")
    assert [c["op"] for c in d["per_step_collectives"] if c["what"] == "records"] == ["gather"] * 3 and d["exchange"]["record_bytes"] == 28
    assert all(c.get("dst", 0) == 0 for c in d["per_step_collectives"])
    root, other = d["ranks"][0], d["ranks"][1]
    assert root["replicated_positions"] == list(range(48, 56)) and other["replicated_positions"] == [] and other["cloud_buffer_bytes"] == 0
    assert root["sharded_positions"] == list(range(0, 48, 4)) and root["launches_per_step"] == 3 + 1
    w = _plan("--gpus", "2", "--scaling", "weak", "--refs", "5", "--workload", "config2")
    assert w["refs_total"] == 10 and w["neighbours"] == 3 and [len(r["sharded_positions"]) for r in w["ranks"]] == [5, 5]


def test_every_baseline_configuration_is_a_workload():
    """BASELINE.json's configurations 2-5 by name (config[0], the two-view CPU case, is a parity test, not a bench line): the plan names the
    configuration's cameras, grid and pair count."""
    want = {"config2": (64, 3, [512, 512], "config[1]"), "config3": (32, 3, [960, 960], "config[2]"),
            "config4": (56, 8, [512, 512], "config[3]"), "config5": (12, 8, [1280, 1280], "config[4]")}
    for name, (refs, k, grid, label) in want.items():
        d = _plan("--gpus", "2", "--workload", name)
        assert (d["refs_total"], d["neighbours"], d["grid"]) == (refs, k, grid) and label in d["workload"], (name, d["workload"])
    assert "194 cameras 1237x822" in _plan("--gpus", "2", "--workload", "config3")["workload"]        # bicycle's image set
