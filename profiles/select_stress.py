"""Stress of the multi-workgroup selection kernel (grid barriers): many calls, several sizes and workgroup counts, each
compared with the single-workgroup kernel - cells, order and the position of the MT19937 stream."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
dens = hb.HipDensifier(dev)
rs = np.random.RandomState(0)
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for it in range(n):
    h, w = [(320, 320), (512, 512), (500, 333), (640, 640), (257, 301)][it % 5]
    M = int(rs.choice([2000, 6000, 10000, 12000]))
    perm = rs.permutation(h * w).astype(np.float64)
    cert = (0.2 + 0.7 * (perm + 0.5) / (h * w)).astype(np.float32).reshape(h, w)
    if it % 3 == 0:
        cert[rs.randint(0, h // 2): h // 2 + 20, :] = 0.0
    t = torch.from_numpy(cert).to(dev)
    res = {}
    for g in (0, int(rs.choice([2, 3, 8, 16, 31, 64]))):
        os.environ["LFD_SELECT_WORKGROUPS"] = str(g)
        dens.reload_env()
        dens.seed_rng(it)
        a = dens.select_samples(t, M).cpu().numpy()
        b = dens.select_samples(t, M).cpu().numpy()
        res[g == 0] = (a, b, dens.rng_state()[1])
    ok = np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1]) and res[True][2] == res[False][2]
    bad += not ok
print("calls compared: %d, mismatches: %d" % (n, bad))
