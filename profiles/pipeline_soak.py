"""Thirty dense_init runs back to back in ONE process (the plugin lives inside LichtFeld Studio for hours): sampled / dense, streamed or not, host / device image
preparation alternating on a 24-camera on-disk scene.  After every run: device memory allocated / reserved by torch (after the run's own empty_cache), the
process's resident set, its threads and open file descriptors.  A leak shows as a column that keeps growing after the first rounds."""
import gc, os, sys, tempfile, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import psutil, torch
import bench_pipeline as bp
from lichtfeld_densification_plugin_amd import densify, synthetic
from lichtfeld_densification_plugin_amd.core import hostenv
hostenv.fit_threads_to_quota()
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp()
synthetic.write_colmap_scene(d, n_cams=24)
args = densify.build_argparser().parse_args(["--scene_root", d, "--images_subdir", "images_4", "--num_refs", "0.8", "--nns_per_ref", "3"])
records, refs, nn, _ = densify.plan_scene(args)
m = synthetic.SyntheticMatcher(records, setting="fast", device=dev)
m.precompute(refs, nn, 3)
proc = psutil.Process()
rows = []
N_RUNS = int(os.environ.get("LFD_SOAK_RUNS", "30"))
for it in range(N_RUNS):
    mode = ("sampled", "dense")[it % 2]
    r = bp.run_once(d, m, mode=mode, device_prep=bool((it // 2) % 2), pack_workers=(4, 16)[(it // 4) % 2], refs_per_launch=(4, 16)[(it // 3) % 2])
    gc.collect()
    torch.cuda.synchronize()
    rows.append((it, mode, round(r["seconds"], 3), torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20, proc.memory_info().rss >> 20,
                 threading.active_count(), proc.num_fds()))
    if N_RUNS <= 30 or it % 10 == 9:
        print("run %3d %-7s %.3f s  torch allocated %5d MiB reserved %5d MiB  rss %6d MiB  threads %3d  fds %3d" % rows[-1], flush=True)
first, last = rows[len(rows) // 2], rows[-1]
grew = {"allocated": last[3] - first[3], "reserved": last[4] - first[4], "rss": last[5] - first[5], "threads": last[6] - first[6], "fds": last[7] - first[7]}
print("growth over the second half of the runs:", grew)
assert grew["allocated"] <= 8 and grew["threads"] <= 0 and grew["fds"] <= 0 and grew["rss"] <= 200, grew
print("SOAK_OK")
