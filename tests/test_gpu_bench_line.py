"""The N = 1 line of bench.py carries what the contract asks for (small shape, seconds): metric / value / unit / config, `roofline` with both
bounds (HBM bytes and vector-issue time), `cpu_baseline`, `parity`, and the round-4 side keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_bench_line_contract():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LFD_BENCH_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2", "--refs", "6", "--preset", "turbo", "--cpu-sample-refs", "2",
           "--parity-refs", "1", "--spinup-s", "0.05"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("triangulated points/sec") and d["unit"] == "points/s" and d["value"] > 0 and d["n_gpus"] == 1
    assert d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "config[1]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"] == "lfd_dense_kernel" and r["kernel_ms"] > 0 and "traffic" in r and "valu" in r and "valu_busy_frac" in r
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "points/s" and "sample" in c
    assert d["parity"]["flipped_out_of_band"] == 0 and d["parity"]["cells"] == 320 * 320
    u = d["unordered_retirement"]
    assert u["kernel"] == "lfd_dense_segments_kernel" and u["kernel_ms"] > 0 and u["survivors"] > 0
    s = d["sampled_mode"]
    assert s["ms_per_reference"] > 0 and s["pipelined_ms_per_reference"] > 0 and s["default_config_ms_per_reference"] > 0 and s["grouped"]["ms_per_reference"] > 0
    assert d["end_to_end"] is None and "unmeasured" in d["end_to_end_note"]
    assert len(s["default_config_passes_ms_per_reference"]) == 5 and s["default_config_ms_per_reference"] == min(s["default_config_passes_ms_per_reference"])
    assert d["host"]["torch_threads"] >= 1 and d["host"]["cpus_visible"] >= d["host"]["torch_threads"]       # threads fitted to the container's quota (core/hostenv.py)
