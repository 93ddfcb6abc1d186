#!/usr/bin/env python3
"""UPSTREAM'S own ``run_dense_pipeline`` on the seeded scenes of tests/fuzz_scenes.py -> tests/golden/g12_pipeline_upstream.npz (development container
only: it imports /root/reference through tests/golden/ref_import.py; nothing of it is copied, the fixture holds inputs' seeds and upstream's OUTPUTS).

    python tests/golden/make_pipeline_fixture.py

tests/golden/check_pipeline_fuzz.py compares the two drivers in one process here; this fixture lets the GPU box - where upstream's code cannot travel -
compare the DEVICE pipeline with what upstream itself computed (tests/test_gpu_pipeline_fuzz.py), and the CPU tier the CPU twin
(tests/test_pipeline_upstream_fixture.py): per scene the survivors (xyz, rgb, err as f32), the processed / matched counters, the progress sequence
(percentages and message heads), the intermediate previews' names and vertex counts - or the error text when upstream raises."""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ref_import import load_reference  # noqa: E402
import fuzz_scenes  # noqa: E402


def main():
    ns = load_reference()
    P = ns.pipeline
    out = {"numpy": np.array(np.__version__), "torch": np.array(torch.__version__), "torch_threads": np.int64(torch.get_num_threads()),
           "n_scenes": np.int64(fuzz_scenes.N_SCENES)}
    total = 0
    for sc in range(fuzz_scenes.N_SCENES):
        with tempfile.TemporaryDirectory() as d:
            cams, refs, nn, table, size, kw = fuzz_scenes.scene(sc, d)
            fm = fuzz_scenes.Table(size[0], size[1], table)
            P.RomaMatcher = lambda device="cpu", mode="outdoor", setting="fast", _fm=fm: _fm
            P.has_cached_romav2_weights = lambda: True
            progress, viz = [], []
            # (upstream's loader hands packages over in completion order with several workers - its run then depends on thread timing and a matcher that
            # replays a table cannot follow it: one worker there; this package's prefetcher is ordered and must give the one-worker result whatever it is set to)
            cfg = ns.config.DensePipelineConfig(output_path=os.path.join(d, "up", "dense.ply"), roma_setting="fast", **dict(kw, pack_workers=1))
            pre = f"s{sc}_"
            try:
                with np.errstate(all="ignore"):
                    res = P.run_dense_pipeline(cams, refs, nn, cfg, progress_callback=lambda p, m: progress.append((round(float(p), 6), m.split(" | ")[0])),
                                               on_sequential_viz=lambda path: viz.append(path))
                out[pre + "xyz"], out[pre + "rgb"], out[pre + "err"] = (np.asarray(a, np.float32) for a in (res.xyz, res.rgb, res.err))
                out[pre + "processed"] = np.int64(res.pairs_processed)
                out[pre + "matched"] = np.int64(fm.calls and sum(len(t) for t in table[:fm.calls]))
                total += int(res.xyz.shape[0])
                names, counts = [], []
                for v in viz:
                    head = open(v, "rb").read().split(b"end_header\n", 1)[0].decode()
                    names.append(os.path.basename(v))
                    counts.append(int([ln for ln in head.split("\n") if ln.startswith("element vertex")][0].split()[-1]))
                out[pre + "previews"] = np.array(json.dumps([names, counts]))
            except Exception as exc:                          # noqa: BLE001 - "No points triangulated": both sides must say so
                out[pre + "error"] = np.array(f"{type(exc).__name__}: {exc}")
            out[pre + "progress"] = np.array(json.dumps(progress))
            out[pre + "cfg"] = np.array(json.dumps(kw))
    path = os.path.join(HERE, "g12_pipeline_upstream.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {fuzz_scenes.N_SCENES} scenes, {total} points, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
