"""lfd_prepare_batch: the tables and per-pair constants of batch i+1 are staged on the context's preparation stream, into the slot the
batch before last used, while the kernels of batch i run.  Back-to-back launches over rotating batches must return exactly what a fresh
context returns for each batch alone (a race between the staging and a running kernel would show as wrong constants)."""
import numpy as np
import pytest
import torch

import lichtfeld_densification_plugin_amd as lfd
from lichtfeld_densification_plugin_amd import synthetic
from lichtfeld_densification_plugin_amd.core import hip_backend as hb

pytestmark = pytest.mark.gpu


def _refs(dev, cams, ids, k, H, W):
    out = []
    for gi, ref in enumerate(ids):
        nbrs = synthetic.ring_neighbours(len(cams), ref, k)
        s = synthetic.synth_reference(cams, ref, nbrs, H, W, W, H, noise_px=0.5, outlier_frac=0.05, channels=2, seed=300 + gi, cert_mode="smooth", device=dev)
        out.append(hb.ReferenceInputs(ref_cam=ref, nbr_cams=nbrs, cert=[s.cert[j] for j in range(k)], warp=[s.warp[j] for j in range(k)], image=s.image))
    return out


@pytest.mark.parametrize("with_prepare", [True, False])
def test_staged_batches_equal_fresh_contexts(with_prepare):
    dev = torch.device("cuda:0")
    H = W = 256
    cams = synthetic.ring_cameras(60, seed=0)
    groups = [_refs(dev, cams, ids, 3, H, W) for ids in ([0, 7, 14, 21], [3, 33, 43], [50, 5, 11, 17, 23], [9, 29])]
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    expected = []
    for g in groups:                                   # every batch alone, on a context of its own
        d = hb.HipDensifier(dev)
        d.upload_cameras(cams)
        o = d.triangulate_dense(hb.PreparedBatch(g, W, H, cameras=cams), params)
        expected.append((o.count, o.xyz.clone(), o.rgb.clone(), o.err.clone(), o.ref_offsets.copy()))
        d.close()
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batches = [hb.PreparedBatch(g, W, H, cameras=cams) for g in groups]
    outs = [hb.OutputBuffers(len(g) * H * W, len(g), 3, dev) for g in groups]
    order = [0, 1, 2, 3, 0, 2, 1, 3, 3, 0, 1, 2] * 4
    rounds = []
    if with_prepare:
        dens.prepare(batches[order[0]], params)
    for i, b in enumerate(order):                      # everything enqueued back to back, nothing read until the end of a round of four
        dens.launch_dense(batches[b], params, outs[b])
        if with_prepare and i + 1 < len(order):
            dens.prepare(batches[order[i + 1]], params)      # staged while the launch above runs
        if i % 4 == 3:
            dens.check_launches()
            for bb in set(order[i - 3:i + 1]):
                r = outs[bb].collect()
                rounds.append((bb, r.count, r.xyz.clone(), r.rgb.clone(), r.err.clone(), r.ref_offsets.copy()))
    assert len(rounds) >= 30
    for bb, n, xyz, rgb, err, offs in rounds:
        e = expected[bb]
        assert n == e[0] and np.array_equal(offs, e[4])
        assert torch.equal(xyz, e[1]) and torch.equal(rgb, e[2]) and torch.equal(err, e[3])
    # the read-back of the constants refers to the batch of the LAST launch, whatever was staged after it
    dens.prepare(batches[0], params)
    F = dens.pair_fundamentals(len(groups[order[-1]]), 3)
    a, b0 = groups[order[-1]][0].ref_cam, groups[order[-1]][0].nbr_cams[0]
    np.testing.assert_array_equal(F[0, 0].astype(np.float32), hb.fundamental_from_world2cam(cams[a].K, cams[a].R, cams[a].t, cams[b0].K, cams[b0].R, cams[b0].t))
    dens.close()


def test_kernel_timing_reports_each_timed_launch_and_changes_nothing():
    """lfd_kernel_timing: the next n dense launches carry a start and a stop event of their own (the kernel's device-side duration);
    the durations come back in launch order, launches beyond n are not timed, results are those of untimed launches, and the
    figure is not larger than what events recorded around the call see."""
    dev = torch.device("cuda:0")
    H = W = 256
    cams = synthetic.ring_cameras(60, seed=0)
    refs = _refs(dev, cams, [0, 7, 14, 21, 28, 35], 3, H, W)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    dens = hb.HipDensifier(dev)
    dens.upload_cameras(cams)
    batch = hb.PreparedBatch(refs, W, H, cameras=cams)
    out = hb.OutputBuffers(len(refs) * H * W, len(refs), 3, dev)
    dens.launch_dense(batch, params, out)
    plain = out.collect()
    want = (plain.count, plain.xyz.clone(), plain.rgb.clone(), plain.err.clone())
    assert dens.dense_kernel_times_ms().size == 0                 # nothing was asked for
    dens.time_dense_kernels(4)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
    for a, b in ev:                                               # six launches back to back, four of them timed
        a.record()
        dens.launch_dense(batch, params, out)
        b.record()
    t = dens.dense_kernel_times_ms()
    assert t.shape == (4,) and np.all(t > 0.0)
    bracket = np.array([a.elapsed_time(b) for a, b in ev[:4]])
    assert np.all(t <= bracket * 1.02 + 0.002), (t, bracket)     # the kernel alone never takes longer than the call's bracket
    assert t.mean() > 0.3 * bracket.mean()
    got = out.collect()
    assert got.count == want[0] and torch.equal(got.xyz, want[1]) and torch.equal(got.rgb, want[2]) and torch.equal(got.err, want[3])
    assert dens.dense_kernel_times_ms().size == 0                 # read once
    dens.launch_dense(batch, params, out)                         # a new series starts at the first event pair
    dens.launch_dense(batch, params, out)
    assert dens.dense_kernel_times_ms().shape == (2,)
    dens.time_dense_kernels(0)
    dens.launch_dense(batch, params, out)
    assert dens.dense_kernel_times_ms().size == 0
    dens.check_launches()
    dens.close()


def test_a_staged_batch_serves_the_indexed_entry_point_too():
    """lfd_triangulate_indexed used to put its selection offsets INTO the cached table blob, so a batch staged by lfd_prepare_batch never
    matched its launch (re-upload in the launch stream, the other slot's tables evicted).  The offsets now travel apart from the cached tables: staged
    or not, the same result - and the slot staged for batch B still serves B after an indexed launch of batch A went through the other slot."""
    dev = torch.device("cuda:0")
    H = W = 128
    cams = synthetic.ring_cameras(60, seed=0)
    ga, gb = _refs(dev, cams, [0, 7, 14], 3, H, W), _refs(dev, cams, [3, 33], 3, H, W)
    params = hb.make_params(lfd.DensePipelineConfig(output_path=""))
    g = torch.Generator().manual_seed(1)
    sel_a = [torch.sort(torch.randperm(H * W, generator=g)[:700]).values for _ in ga]
    sel_b = [torch.sort(torch.randperm(H * W, generator=g)[:500]).values for _ in gb]

    def run(prepare):
        d = hb.HipDensifier(dev)
        d.upload_cameras(cams)
        ba, bb = hb.PreparedBatch(ga, W, H, cameras=cams), hb.PreparedBatch(gb, W, H, cameras=cams)
        res = []
        for rep in range(3):
            for batch, sels in ((ba, sel_a), (bb, sel_b)):
                if prepare:
                    d.prepare(batch, params)
                offs = np.concatenate([[0], np.cumsum([int(s.numel()) for s in sels])]).tolist()
                if rep == 1:                      # other offsets for the same tables: only the offsets change
                    sels = [s[:300] for s in sels]
                    offs = np.concatenate([[0], np.cumsum([300] * len(sels))]).tolist()
                o = d.triangulate_indexed(batch, params, torch.cat(sels).to(dev), offs)
                res.append((o.count, o.xyz.clone(), o.rgb.clone(), o.err.clone(), o.ref_offsets.copy()))
        d.close()
        return res
    plain, staged = run(False), run(True)
    assert len(plain) == 6 and plain[0][0] > 500
    for p, s in zip(plain, staged):
        assert p[0] == s[0] and np.array_equal(p[4], s[4]) and torch.equal(p[1], s[1]) and torch.equal(p[2], s[2]) and torch.equal(p[3], s[3])
