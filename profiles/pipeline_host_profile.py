import sys, os, time, cProfile, pstats, io, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench_pipeline as bp
from lichtfeld_densification_plugin_amd import densify, synthetic
from lichtfeld_densification_plugin_amd.core import hostenv
hostenv.fit_threads_to_quota()
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp()
synthetic.write_colmap_scene(d, n_cams=185)
args = densify.build_argparser().parse_args(["--scene_root", d, "--images_subdir", "images_4", "--num_refs", "0.8", "--nns_per_ref", "3"])
records, refs, nn, _ = densify.plan_scene(args)
m = synthetic.SyntheticMatcher(records, setting="fast", device=dev)
m.precompute(refs, nn, 3)
for mode in ("dense", "sampled"):
    bp.run_once(d, m, mode=mode, device_prep=True, pack_workers=16)
    pr = cProfile.Profile()
    pr.enable()
    r = bp.run_once(d, m, mode=mode, device_prep=True, pack_workers=16)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
    print(mode, r["seconds"], r["stage_seconds"])
    print(s.getvalue()[:7000])
