"""Device-side record packing (N1) against upstream's writers: byte-exact."""
import numpy as np
import pytest
import torch

from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from lichtfeld_densification_plugin_amd.core import writers
from helpers import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dens():
    d = hb.HipDensifier(torch.device("cuda:0"))
    yield d
    d.close()


def test_packed_files_equal_upstream_golden(g5, dens, tmp_path):
    dev = dens.device
    xyz, rgb, err = (torch.from_numpy(g5[k]).to(dev) for k in ("xyz", "rgb", "err"))
    ply = dens.pack_ply(xyz, rgb).cpu().numpy().tobytes()
    p3d = dens.pack_points3d(xyz, rgb, err).cpu().numpy().tobytes()
    np.testing.assert_array_equal(dens.quantise_rgb(rgb).cpu().numpy(), g5["rgb_u8"])
    p1, p2 = tmp_path / "a.ply", tmp_path / "a.bin"
    writers.write_ply_packed(str(p1), 5, ply)
    writers.write_points3D_bin_packed(str(p2), 5, p3d)
    assert p1.read_bytes() == g5["ply"].tobytes()
    assert p2.read_bytes() == g5["points3d_bin"].tobytes()


@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 257, 100003])
def test_packing_matches_oracle_writer_for_ragged_sizes(dens, n):
    rs = np.random.RandomState(n)
    xyz = rs.normal(0, 5, (n, 3)).astype(np.float32)
    rgb = rs.uniform(-0.1, 1.1, (n, 3)).astype(np.float32)
    if n > 10:
        rgb[:6] = np.array([[0.5 / 255, 1.5 / 255, 2.5 / 255], [254.5 / 255, 255.5 / 255, 0.0], [np.nan, np.inf, -np.inf],
                            [1.0, 0.0, 0.5], [0.49803922, 0.5019608, 0.0019607844], [1e-8, -1e-8, 0.99999994]], np.float32)
    err = rs.uniform(0, 2, n).astype(np.float32)
    dev = dens.device
    tx, tc, te = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev), torch.from_numpy(err).to(dev)
    with np.errstate(invalid="ignore"):
        u8 = orc.to_uint8_rgb(rgb)
    head = orc.ply_bytes(xyz, u8)[:-15 * n] if n else orc.ply_bytes(xyz, u8)
    assert head + dens.pack_ply(tx, tc).cpu().numpy().tobytes() == orc.ply_bytes(xyz, u8)
    assert np.uint64(n).tobytes() + dens.pack_points3d(tx, tc, te).cpu().numpy().tobytes() == orc.points3d_bin_bytes(xyz, u8, err)
    assert np.uint64(n).tobytes() + dens.pack_points3d(tx, tc, None).cpu().numpy().tobytes() == orc.points3d_bin_bytes(xyz, u8, None)


@pytest.mark.gpu
def test_copy_segments_places_byte_ranges_in_one_launch():
    """lfd_copy_segments: arbitrary byte offsets and lengths (15-byte records: nothing is aligned), empty segments, more segments than one
    launch's argument block holds, a segment longer than one workgroup's chunk - against the same copies done one by one"""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(4)
    src = torch.from_numpy(rs.randint(0, 256, size=3_000_000, dtype=np.uint8)).to(dev)
    for n_seg in (1, 7, 96, 97, 250):
        lens = rs.randint(0, 20000, size=n_seg).astype(np.int64)
        lens[0] = 150_001                                            # several chunks, odd length
        if n_seg > 3:
            lens[3] = 0
        s_off = np.sort(rs.choice(3_000_000 - 160_000, size=n_seg, replace=False)).astype(np.int64)
        gaps = rs.randint(0, 23, size=n_seg).astype(np.int64)       # destinations: packed with small odd gaps
        d_off = np.cumsum(gaps + np.concatenate([[0], lens[:-1]]))
        total = int(d_off[-1] + lens[-1]) + 5
        dst = torch.full((total,), 0xEE, dtype=torch.uint8, device=dev)
        exp = np.full(total, 0xEE, np.uint8)
        host = src.cpu().numpy()
        for s_, d_, n_ in zip(s_off, d_off, lens):
            exp[d_:d_ + n_] = host[s_:s_ + n_]
        hb.copy_segments(src, dst, np.stack([s_off, d_off, lens], 1))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(dst.cpu().numpy(), exp)
    # typed tensors (28-byte rows), a destination that is a view into a larger buffer
    rows = torch.arange(7 * 1000, dtype=torch.float32, device=dev).reshape(1000, 7)
    big = torch.zeros((2000, 7), dtype=torch.float32, device=dev)
    hb.copy_segments(rows, big[500:1500], [(100 * 28, 0, 50 * 28), (0, 50 * 28, 100 * 28)])
    torch.cuda.synchronize()
    assert torch.equal(big[500:550], rows[100:150]) and torch.equal(big[550:650], rows[0:100]) and float(big[:500].abs().sum()) == 0.0 and float(big[650:].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        hb.copy_segments(rows, big[500:1500], [(0, 999 * 28, 2 * 28)])            # leaves the destination view


@pytest.mark.gpu
def test_device_packers_equal_upstreams_writers_on_adversarial_values():
    """lfd_pack_ply / lfd_pack_points3d / lfd_quantise_rgb against the oracle's writers (themselves equal to upstream's, byte for byte:
    tests/golden/g9_oracle_fuzz.json) on seeded arrays with every value where the colour quantisation can go wrong - exact halves k + 0.5 (round half to
    even), values just below and above them, negatives, values above 1, NaN / Inf colours and positions - and on the sizes 0, 1, 255, 256, 257, 100 003"""
    dev = torch.device("cuda:0")
    dens = hb.HipDensifier(dev)
    rs = np.random.RandomState(8)
    for n in (0, 1, 255, 256, 257, 100_003):
        xyz = rs.normal(0, 50, (n, 3)).astype(np.float32)
        rgb = rs.uniform(-0.1, 1.1, (n, 3)).astype(np.float32)
        err = rs.uniform(0, 3, (n,)).astype(np.float32)
        if n >= 255:
            halves = ((np.arange(0, 255) + 0.5) / 255.0).astype(np.float32)
            rgb[:255, 0] = halves
            rgb[:255, 1] = np.nextafter(halves, np.float32(2))
            rgb[:255, 2] = np.nextafter(halves, np.float32(-1))
            rgb[3] = [np.nan, np.inf, -np.inf]
            xyz[5] = [np.nan, -np.inf, 1e38]
            err[7] = np.inf
        tx, tc, te = (torch.from_numpy(a).to(dev) for a in (xyz, rgb, err))
        with np.errstate(all="ignore"):
            u8 = orc.to_uint8_rgb(rgb)
            want_ply = orc.ply_bytes(xyz, u8).split(b"end_header\n", 1)[1]
            want_p3d = orc.points3d_bin_bytes(xyz, u8, err)[8:]                     # behind the u64 count
        assert dens.quantise_rgb(tc).cpu().numpy().tobytes() == u8.tobytes()
        assert dens.pack_ply(tx, tc).cpu().numpy().tobytes() == want_ply
        assert dens.pack_points3d(tx, tc, te, id_base=0).cpu().numpy().tobytes() == want_p3d
    dens.close()
