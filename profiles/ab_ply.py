#!/usr/bin/env python3
"""A/B of the PLY-record kernels (lfd_dense_ply_kernel, lfd_dense_ply_segments_kernel) over build/variants/*.so, alternating passes on one lease:
python profiles/ab_ply.py [--passes 3]  - the bench's own `ply_output` leg (100 launches each, the kernel's start / stop events)."""
import argparse, glob, json, os, subprocess, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--passes", type=int, default=3); a = ap.parse_args()
libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so")))
res = {os.path.basename(l)[:-3]: ([], []) for l in libs}
for p in range(a.passes):
    for lib in libs:
        name = os.path.basename(lib)[:-3]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-refs", "0", "--parity-refs", "0", "--pipeline-cams", "0", "--steps", "50"],
                             env=dict(os.environ, LFD_DENSIFY_LIB=lib), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode or not line:
            print(name, "FAILED", out.stderr[-300:]); continue
        d = json.loads(line[0])["ply_output"]
        res[name][0].append(d["kernel_ms"]); res[name][1].append(d["unordered"]["kernel_ms"])
        print(f"pass {p} {name:20s} ordered {d['kernel_ms']:.4f}  unordered {d['unordered']['kernel_ms']:.4f}", flush=True)
for name, (o, u) in res.items():
    if o:
        print(f"{name:20s} ordered {np.mean(o):.4f} +- {np.std(o):.4f}   unordered {np.mean(u):.4f} +- {np.std(u):.4f}")
