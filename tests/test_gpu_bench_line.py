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
           "--parity-refs", "1", "--spinup-s", "0.05", "--pipeline-cams", "8", "--pipeline-size", "320x208", "--pipeline-latency-ms", "1"]
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
    pw = r["power"]          # the package's power and the shader clock beside back-to-back launches (sysfs), or the reason they could not be read
    assert ("note" in pw) and (("package_W" not in pw) or (pw["package_W"] > 50 and pw["cap_W"] >= pw["package_W"] * 0.5 and pw["sclk_MHz"] > 100 and pw["samples"] > 5))
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "points/s" and "sample" in c
    assert d["parity"]["flipped_out_of_band"] == 0 and d["parity"]["cells"] == 320 * 320
    u = d["unordered_retirement"]
    assert u["kernel"] == "lfd_dense_segments_kernel" and u["kernel_ms"] > 0 and u["survivors"] > 0
    po = d["ply_output"]
    assert po["kernel"] == "lfd_dense_ply_kernel" and po["unordered"]["kernel"] == "lfd_dense_ply_segments_kernel" and po["unordered"]["survivors"] == po["survivors"] == u["survivors"]
    dm = d["default_mode"]                                   # the user-visible default beside the dense headline
    assert dm["ms_per_reference"] > 0 and abs(dm["refs_per_s"] - 1e3 / dm["ms_per_reference"]) < 1e-6 * dm["refs_per_s"] and dm["pairs_per_s"] > dm["refs_per_s"]
    one = dm["one_reference_per_call"]                       # what a run with intermediate previews uses; the automatic default groups references on the one stream
    assert "refs_per_launch=0" in dm["mode"] and one["ms_per_reference"] > dm["ms_per_reference"] > 0 and dm["ms_per_reference_device_sums"] > 0
    assert d["rccl_ranks"] == 0 and d["collective_backend"] is None
    rp = d["cpu_baseline"]["reference_python"]               # upstream's own code, timed in the development container (tests/golden/g11_reference_timing.json)
    assert rp["cores"] >= 1 and rp["ms_per_reference"] > 10 and rp["run_dense_pipeline"]["pack_workers_1"]["references"] > 100
    pl = d["pipeline"]                                       # the end-to-end leg on a small scene (--pipeline-cams 8)
    assert pl["scene"]["cameras"] == 8 and pl["sampled"]["device_prep"]["points"] > 0 and pl["dense"]["device_prep"]["d2h_bytes"] == 15 * pl["dense"]["device_prep"]["points"]
    s = d["sampled_mode"]
    assert s["ms_per_reference"] > 0 and s["pipelined_ms_per_reference"] > 0 and s["default_config_ms_per_reference"] > 0 and s["grouped"]["ms_per_reference"] > 0
    assert s["chained"]["ms_per_reference"] > 0 and s["chained"]["default_config_ms_per_reference"] == dm["ms_per_reference"]
    assert d["end_to_end"] is None and "unmeasured" in d["end_to_end_note"]
    assert len(s["default_config_passes_ms_per_reference"]) == 5 and s["default_config_ms_per_reference"] == min(s["default_config_passes_ms_per_reference"])
    assert d["host"]["torch_threads"] >= 1 and d["host"]["cpus_visible"] >= d["host"]["torch_threads"]       # threads fitted to the container's quota (core/hostenv.py)
