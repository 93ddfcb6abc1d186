"""Soak of the dense kernel's ordered retirement and of the staged batch preparation: many back-to-back launches on the bench workload and
on a ragged small one - three distinct batches in rotation (the references in another order), each staged by lfd_prepare_batch on the
preparation stream while the launch before it runs - every sampled launch's survivor count, reference offsets and checksums of the records
compared with the first launch of the same batch.  python profiles/dense_soak.py [launches]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
for (n_refs, H, W, k) in ((64, 512, 512, 3), (7, 90, 126, 2), (24, 320, 320, 4)):
    cams = synthetic.ring_cameras(64, seed=0)
    refs = []
    for r in range(n_refs):
        nb = synthetic.ring_neighbours(64, r, k)
        s = synthetic.synth_reference(cams, r, nb, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=r, device=dev)
        refs.append(hb.ReferenceInputs(ref_cam=r, nbr_cams=nb, cert=[s.cert[j] for j in range(k)], warp=[s.warp[j] for j in range(k)], image=s.image))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batches = [hb.PreparedBatch(refs[j:] + refs[:j], W, H) for j in range(3)]
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    out = hb.OutputBuffers(n_refs * H * W, n_refs, k, dev, with_cell=True)
    first = {}
    t0 = time.time()
    dens.prepare(batches[0], params)
    for i in range(n_launch):
        dens.launch_dense(batches[i % 3], params, out)
        sample = i % 50 in (0, 1, 2) or i == n_launch - 1
        if not sample and i + 1 < n_launch:
            dens.prepare(batches[(i + 1) % 3], params)          # staged beside the launch above
        if sample:
            dens.check_launches()
            res = out.collect()
            sig = (res.count, tuple(int(v) for v in res.ref_offsets), float(res.xyz.double().sum().item()), int(res.cell.long().sum().item()))
            assert first.setdefault(i % 3, sig) == sig, (i, sig[:1], first[i % 3][:1])
    torch.cuda.synchronize()
    print(f"{n_refs} refs x {k} nbrs x {H}x{W}: {n_launch} launches over 3 rotating batches, {first[0][0]} survivors each time, {time.time() - t0:.1f} s")
    # round 4: the same soak over the other two forms of the kernel, INTERLEAVED with the ordered one on one context (they share the tile-state
    # words, the ticket counters and the per-reference cursors): unordered retirement + the table consumers, and the kernel-written PLY records -
    # every sampled launch against the ordered kernel's result of the same batch
    ref_ply = {j: dens.pack_ply(*(lambda r: (r.xyz, r.rgb))(dens.triangulate_dense(batches[j], params))).clone() for j in range(3)}
    tpr = (H * W + 1023) // 1024
    table = torch.zeros((n_refs * tpr, 2), dtype=torch.int32, device=dev)
    counts = torch.zeros((n_refs,), dtype=torch.int64, device=dev)
    rec = torch.empty((n_refs * H * W * 15,), dtype=torch.uint8, device=dev)
    offs = torch.zeros((n_refs + 1,), dtype=torch.int64, device=dev)
    t1 = time.time()
    n2 = max(n_launch // 2, 30)
    for i in range(n2):
        j = i % 3
        form = i % 4
        if form == 0:
            dens.launch_dense(batches[j], params, out)
        elif form in (1, 3):
            dens.launch_dense_segments(batches[j], params, out, table, counts)
        else:
            dens.launch_dense_ply(batches[j], params, rec, offs)
        if i % 40 in (1, 2, 3) or i == n2 - 1:
            dens.check_launches()
            if form in (1, 3):
                body, _o = dens.pack_ply_segments(hb.SegmentedOutput(out, table, counts, n_refs, H, W, k))
                assert torch.equal(body, ref_ply[j]), (i, "segments")
            elif form == 2:
                n = int(offs[-1].item())
                assert torch.equal(rec[:n * 15], ref_ply[j]), (i, "ply")
    torch.cuda.synchronize()
    print(f"   + {n2} launches alternating ordered / unordered / PLY-record forms on the same context: every sampled result = the ordered kernel's bytes, {time.time() - t1:.1f} s")
    dens.close()
