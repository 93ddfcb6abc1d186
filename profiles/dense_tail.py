"""Where the heavy tail of the dense kernel's front end comes from (stamps of a -DLFD_DENSE_TIMING build, see dense_phases.py):
python profiles/dense_tail.py <file> [clock_mhz].  The look-back makes every tile wait for the SLOWEST of the tiles before it, so the tails of the
phases before a tile's count - not their means - set that wait."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 12).astype(np.int64)
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2100.0
idx = np.arange(a.shape[0])
ok = (a[:, 0, 0] > 0) & (a[:, 0, 11] > 0)
a, idx = a[ok], idx[ok]
held = (a[:, 0, 0] & 1).astype(bool)
d0 = np.diff(a[:, 0, :], axis=1) / mhz
pc = lambda v: " ".join(f"p{q} {np.percentile(v, q):6.2f}" for q in (50, 75, 90, 95, 99, 99.9))
to_count = (a[:, 0, 6] - a[:, 0, 0]) / mhz          # entry -> wave counts exchanged (the count is published right behind it)
print(f"{a.shape[0]} tiles; entry -> count known: mean {to_count.mean():.2f} us  {pc(to_count)}")
for name, v in (("ticket + constants requested (+ barrier)", d0[:, 0]), ("constants staged (+ barrier)", d0[:, 1]), ("certainty planes", d0[:, 2]), ("winner's warp", d0[:, 3]),
                ("geometry", d0[:, 4]), ("ballots + barrier", d0[:, 5])):
    print(f"  {name:42s} mean {v.mean():6.2f}  {pc(v)}")
print(f"guess held: {held.mean() * 100:.1f} %;  entry -> count, guess held: mean {to_count[held].mean():.2f} {pc(to_count[held])}")
print(f"                     entry -> count, guess wrong: mean {to_count[~held].mean():.2f} {pc(to_count[~held])}")
skew = (a[:, 1, 0] - a[:, 0, 0]) / mhz             # wave 1 enters the kernel this much after wave 0 (same CU, same clock)
print(f"wave 1 enters {skew.mean():.2f} us after wave 0 on average ({pc(np.abs(skew))} of |skew|)")
gen = idx // 2048
for g in range(int(gen.max()) + 1):
    m = gen == g
    if m.sum():
        print(f"  generation {g}: tiles {m.sum():5d}  ticket phase mean {d0[m, 0].mean():5.2f} p95 {np.percentile(d0[m, 0], 95):5.2f} p99 {np.percentile(d0[m, 0], 99):5.2f}   entry -> count p50 {np.median(to_count[m]):5.2f} p99 {np.percentile(to_count[m], 99):5.2f}")
# the slowest tiles: which phase made them slow?
slow = to_count > np.percentile(to_count, 99)
print("the slowest 1 % of the tiles (entry -> count), mean per phase against all tiles:")
for i, name in enumerate(("ticket", "constants", "certainty", "warp", "geometry", "ballots")):
    print(f"  {name:10s} {d0[slow, i].mean():6.2f} us  (all: {d0[:, i].mean():5.2f})")
