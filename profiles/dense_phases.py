"""Per-phase latency of the dense kernel from the stamps of a -DLFD_DENSE_TIMING build (LFD_DENSE_TIMING=<file> in the
environment): python profiles/dense_phases.py <file> [clock_mhz].  Stamps are shader-clock readings of lane 0 of waves 0 and 1
of every tile: 0 entry, 1 ticket known, 2 constants staged, 3 arg-max done, 4 warps parked, 5 geometry done, 6 wave counts
exchanged, 7 order map complete, 8 look-back (wave 0) / colours (wave 1) done, 9 barrier passed, 10 stores issued, 11 stores retired."""
import sys
import numpy as np

NAMES = ["ticket (atomic + barrier)", "constants -> LDS (+ barrier)", "certainty loads + arg-max", "warp loads + park",
         "geometry (4 cells)", "ballots + barrier", "order map + barrier", "look-back | colours", "barrier (prefix known)",
         "stores issued", "stores retired"]
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 12).astype(np.int64)
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2100.0     # readcyclecounter = s_memtime = shader clock (about 2.1 GHz under this load)
ok = (a[:, 0, 0] > 0) & (a[:, 0, 11] > 0)
a = a[ok]
print(f"front end: the guessed reference held for {np.mean(a[:, 0, 0] & 1) * 100:.1f} % of the tiles (ticket == workgroup index for {np.mean((a[:, 0, 0] >> 1) & 1) * 100:.1f} %)")
print(f"{a.shape[0]} tiles; tile life (entry -> stores retired), wave 0: mean {np.mean(a[:, 0, 11] - a[:, 0, 0]) / mhz:.2f} us, "
      f"p50 {np.median(a[:, 0, 11] - a[:, 0, 0]) / mhz:.2f}, p95 {np.percentile(a[:, 0, 11] - a[:, 0, 0], 95) / mhz:.2f}")
for w in (0, 1):
    d = np.diff(a[:, w, :], axis=1) / mhz
    print(f"wave {w}:")
    for i, n in enumerate(NAMES):
        print(f"  {n:34s} mean {d[:, i].mean():7.2f} us   p50 {np.median(d[:, i]):7.2f}   p95 {np.percentile(d[:, i], 95):7.2f}")
# (the counters of different XCDs are not synchronised: only differences inside one wave mean anything)
