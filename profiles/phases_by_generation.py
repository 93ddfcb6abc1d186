"""Look-back wait and tile life by generation (tile index // resident workgroups) from the stamps of a -DLFD_DENSE_TIMING build:
python profiles/phases_by_generation.py <file> [resident_workgroups] [clock_mhz]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 12).astype(np.int64)
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1792
mhz = float(sys.argv[3]) if len(sys.argv) > 3 else 2100.0
n = a.shape[0]
for g in range((n + G - 1) // G):
    s = a[g * G:(g + 1) * G, 0, :]
    s = s[(s[:, 0] > 0) & (s[:, 11] > 0)]
    if not len(s): continue
    d = np.diff(s, axis=1) / mhz
    print(f"gen {g:2d}: tiles {len(s):5d}  life {np.mean(s[:, 11] - s[:, 0]) / mhz:6.1f} us  front {d[:, 0:4].sum(1).mean():5.1f}  geometry {d[:, 4].mean():5.1f}  "
          f"look-back {d[:, 7].mean():5.1f} (p95 {np.percentile(d[:, 7], 95):5.1f})  end {d[:, 8:11].sum(1).mean():4.1f}")
