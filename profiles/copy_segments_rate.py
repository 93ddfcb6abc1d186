#!/usr/bin/env python3
"""lfd_copy_segments: bytes moved per second for the exchange's placement step (56 references' 15-byte records, ~3.6 MB each, odd byte offsets)
against torch's own device copy of the same bytes and against one tensor assignment per segment."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lichtfeld_densification_plugin_amd.core import hip_backend as hb  # noqa: E402

dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for n_seg, rows in ((56, 238_000), (8, 1_667_000), (448, 29_750)):
    lens = (rs.randint(int(rows * 0.9), int(rows * 1.1), size=n_seg) * 15).astype(np.int64)
    total = int(lens.sum())
    src = torch.empty((total + 64 * n_seg + 64,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    s_off = np.cumsum(np.concatenate([[0], lens[:-1] + 15 * rs.randint(0, 4, size=n_seg - 1)]))          # padded blocks: gaps between the segments
    d_off = np.concatenate([[0], np.cumsum(lens[:-1])]) + 7                                              # packed, starting at an odd byte
    segs = np.stack([s_off, d_off, lens], 1)

    def t(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def one_by_one():
        for s_, d_, n_ in segs:
            dst[d_:d_ + n_] = src[s_:s_ + n_]
    t_seg = t(lambda: hb.copy_segments(src, dst, segs))
    t_torch = t(lambda: dst[:total].copy_(src[:total]))
    t_each = t(one_by_one, reps=5)
    print(f"{n_seg:4d} segments, {total / 1e6:7.1f} MB: lfd_copy_segments {t_seg * 1e3:.3f} ms ({2 * total / t_seg / 1e12:.2f} TB/s read+write) | one torch copy of the same bytes "
          f"{t_torch * 1e3:.3f} ms ({2 * total / t_torch / 1e12:.2f} TB/s) | one tensor assignment per segment {t_each * 1e3:.3f} ms")
