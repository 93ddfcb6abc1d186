#!/usr/bin/env python3
"""A/B of kernel builds on ONE lease (VERDICT r3 "small" 8): every build/variants/*.so is timed `--passes` times, ALTERNATING (a b c a b c ...), with the
bench's own sustained loop and the dense kernel's start / stop events (`bench.py --light --cpu-sample-refs 0 --parity-refs 0 --steps N`); prints every
pass and, per variant, mean +- standard deviation, min, max - differences below the spread are not differences.
Usage (on the GPU box): python profiles/variants_run.py [--passes 5] [--steps 300] [bench args ...]"""
import argparse
import glob
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--steps", type=int, default=300)
    args, extra = ap.parse_known_args()
    libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so")))
    if not libs:
        raise SystemExit("no build/variants/*.so (profiles/variants.sh build ...)")
    res = {os.path.basename(l)[:-3]: [] for l in libs}
    surv = {}
    for p in range(args.passes):
        for lib in libs:
            name = os.path.basename(lib)[:-3]
            env = dict(os.environ, LFD_DENSIFY_LIB=lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--light", "--cpu-sample-refs", "0", "--parity-refs", "0", "--steps", str(args.steps)] + extra,
                                 env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if out.returncode != 0 or not line:
                print(f"pass {p} {name}: FAILED {out.stderr[-300:]}", flush=True)
                continue
            d = json.loads(line[0])
            res[name].append(d["roofline"]["kernel_ms"])
            surv[name] = d["survivor_fraction"]
            print(f"pass {p} {name:28s} kernel_ms {d['roofline']['kernel_ms']:.4f}  step {d['ms_per_step']:.4f}  surv {d['survivor_fraction']:.6f}", flush=True)
    base = None
    for name, v in res.items():
        if not v:
            continue
        m = float(np.mean(v))
        base = m if base is None else base
        print(f"{name:28s} mean {m:.4f} ms +- {np.std(v):.4f} (min {np.min(v):.4f} max {np.max(v):.4f}, {len(v)} passes)  {m / base - 1.0:+.2%} vs first  surv {surv.get(name)}")


if __name__ == "__main__":
    main()
