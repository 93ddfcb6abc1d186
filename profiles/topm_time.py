import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
dev = torch.device("cuda:0")
dens = hb.HipDensifier(dev)
g = torch.Generator(device="cpu"); g.manual_seed(0)
for (h, w, M) in ((320, 320, 12000), (512, 512, 10000), (1280, 1280, 16000)):
    cert = (0.2 + 0.7 * torch.rand((h, w), generator=g)).to(dev)
    for _ in range(3): dens.select_top_m(cert, M)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sel = dens.select_top_m(cert, M)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("top-M %dx%d M=%d: %.3f ms per call (incl. count read-back), %d selected" % (h, w, M, dt * 1e3, sel.numel()))
