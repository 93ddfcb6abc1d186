"""N3 image preparation: Pillow's BILINEAR / NEAREST resampling restated (oracle/image_oracle.py) and run on the device
(csrc/lfd_image.hip), against golden g7 = what upstream's own functions (core/image_utils.py load_rgb_resized /
load_mask_resized_np / apply_mask_to_rgb) return for synthetic images."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from lichtfeld_densification_plugin_amd.core import hip_backend as hb
from oracle import image_oracle as io


@pytest.fixture(scope="module")
def g7():
    return load_golden("g7_image_prep.npz")


def _cases(g7):
    return [str(n) for n in g7["names"]]


def test_oracle_reproduces_upstream_image_prep(g7):
    for name in _cases(g7):
        pre = name + "_"
        size = tuple(int(v) for v in g7[pre + "size"])
        np.testing.assert_array_equal(io.resize_bilinear_u8(g7[pre + "image"], size), g7[pre + "rgb"])
        m01 = io.mask01_resized(g7[pre + "mask"], size)
        np.testing.assert_array_equal(m01, g7[pre + "mask01"])
        np.testing.assert_array_equal(io.mask01_resized(g7[pre + "mask"], size, threshold=0.3, invert=True), g7[pre + "mask01_inv03"])
        np.testing.assert_array_equal(io.black_out(g7[pre + "rgb"], m01), g7[pre + "blacked"])


def test_oracle_equals_pillow_on_random_sizes():
    """The restatement against the library itself (Pillow is what upstream calls), sizes up to the garden / bicycle images."""
    from PIL import Image
    rs = np.random.RandomState(3)
    for h, w, wo, ho in [(37, 53, 31, 29), (29, 31, 53, 37), (840, 1297, 512, 512), (100, 100, 100, 37), (17, 400, 9, 17),
                         # Pillow's Image.resize resamples VERTICALLY FIRST when the image is more than 100 times taller than wide and shrinks
                         # vertically (PIL/Image.py): both sides of that rule
                         (1201, 12, 53, 37), (1200, 12, 53, 37), (1201, 12, 12, 37), (1300, 12, 30, 1301), (301, 3, 3, 300), (997, 3, 31, 31)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        np.testing.assert_array_equal(io.resize_bilinear_u8(img, (wo, ho)), np.asarray(Image.fromarray(img).resize((wo, ho), Image.BILINEAR)))
        m = rs.randint(0, 256, (h, w)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(m, mode="L").resize((wo, ho), Image.NEAREST))
        np.testing.assert_array_equal(m[io.nearest_indices(h, ho)][:, io.nearest_indices(w, wo)] if (w, h) != (wo, ho) else m, ref)


def test_library_tables_equal_the_oracle_tables():
    """The fixed-point coefficient and index tables the kernels use are built by the library's host code in f64 like Pillow's."""
    for a, b in [(97, 64), (1297, 512), (1237, 640), (840, 512), (50, 80), (64, 64), (3, 1), (1, 5)]:
        bo, ko, _ = io.bilinear_coeffs(a, b)
        bd, kd = hb.host_resize_tables(a, b)
        np.testing.assert_array_equal(bd, bo)
        np.testing.assert_array_equal(kd, ko)
        np.testing.assert_array_equal(hb.host_nearest_indices(a, b), io.nearest_indices(a, b))
    np.testing.assert_array_equal(io.mask_threshold_lut(0.5), (np.arange(256) >= 128).astype(np.uint8))


@pytest.mark.gpu
def test_device_image_prep_is_byte_exact(g7):
    dev = torch.device("cuda:0")
    dens = hb.HipDensifier(dev)
    for name in _cases(g7):
        pre = name + "_"
        size = tuple(int(v) for v in g7[pre + "size"])
        img = torch.from_numpy(g7[pre + "image"]).to(dev)
        mask = torch.from_numpy(g7[pre + "mask"]).to(dev)
        rgb = dens.prepare_image(img, size)
        np.testing.assert_array_equal(rgb.cpu().numpy(), g7[pre + "rgb"])
        m01 = dens.prepare_mask(mask, size)
        np.testing.assert_array_equal(m01.cpu().numpy(), g7[pre + "mask01"])
        np.testing.assert_array_equal(dens.prepare_mask(mask, size, threshold=0.3, invert=True).cpu().numpy(), g7[pre + "mask01_inv03"])
        np.testing.assert_array_equal(dens.prepare_image(img, size, m01).cpu().numpy(), g7[pre + "blacked"])
    # full-size images (garden 1297x840 -> fast 512^2, bicycle 1237x822 -> high 640^2) against the oracle, alternating sizes on one context
    rs = np.random.RandomState(5)
    for h, w, wo, ho in [(840, 1297, 512, 512), (822, 1237, 640, 640), (840, 1297, 512, 512), (512, 512, 512, 512)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = dens.prepare_image(torch.from_numpy(img).to(dev), (wo, ho)).cpu().numpy()
        np.testing.assert_array_equal(got, io.resize_bilinear_u8(img, (wo, ho)))
        m = rs.randint(0, 256, (h, w)).astype(np.uint8)
        np.testing.assert_array_equal(dens.prepare_mask(torch.from_numpy(m).to(dev), (wo, ho)).cpu().numpy(), io.mask01_resized(m, (wo, ho)))
    dens.close()


def test_decode_cache_is_bounded_by_bytes(tmp_path, monkeypatch):
    """The caches of DECODED full-resolution files (device_image_prep) hold at most DECODE_CACHE_BYTES of arrays, least recently used
    first out - not 512 entries of whatever size (a 24 MP photograph is 72 MB)."""
    from PIL import Image
    from lichtfeld_densification_plugin_amd.core import image_io
    image_io.decode_rgb_u8.cache_clear()
    paths = []
    rs = np.random.RandomState(0)
    for i in range(6):
        p = os.path.join(str(tmp_path), f"d{i}.png")
        Image.fromarray(rs.randint(0, 256, (100, 120, 3)).astype(np.uint8)).save(p)
        paths.append(p)
    one = 100 * 120 * 3
    monkeypatch.setattr(image_io, "DECODE_CACHE_BYTES", int(3.5 * one))
    arrs = [image_io.decode_rgb_u8(p) for p in paths]
    info = image_io.decode_rgb_u8.cache_info()
    assert info["entries"] == 3 and info["bytes"] == 3 * one and info["misses"] == 6
    assert image_io.decode_rgb_u8(paths[5]) is arrs[5] and image_io.decode_rgb_u8.cache_info()["hits"] == 1      # most recent: still there
    again = image_io.decode_rgb_u8(paths[0])                                                                       # evicted: decoded again
    assert again is not arrs[0] and np.array_equal(again, arrs[0]) and not again.flags.writeable
    monkeypatch.setattr(image_io, "DECODE_CACHE_BYTES", one // 2)                                                  # larger than the budget: not cached
    image_io.decode_rgb_u8.cache_clear()
    image_io.decode_rgb_u8(paths[1])
    assert image_io.decode_rgb_u8.cache_info()["entries"] == 0
    image_io.decode_rgb_u8.cache_clear()


@pytest.mark.gpu
def test_device_image_prep_equals_pillow_on_random_sizes():
    """The device resize / mask kernels against the LIBRARY upstream calls (Pillow itself, not the restatement): 40 seeded size pairs - odd sizes,
    up- and down-scaling (down to 1 pixel, up 8x), extreme aspect ratios, identity - images, masks with both thresholds, and the black-out."""
    from PIL import Image
    dev = torch.device("cuda:0")
    dens = hb.HipDensifier(dev)
    rs = np.random.RandomState(11)
    sizes = [(1, 1, 1, 1), (1, 7, 5, 1), (3, 3, 24, 24), (64, 64, 64, 64), (997, 3, 31, 31), (2, 1300, 640, 2),
             (1201, 12, 53, 37), (1200, 12, 53, 37), (1201, 12, 12, 37), (1300, 12, 30, 1301), (301, 3, 3, 300), (2001, 20, 7, 13)]      # Pillow's vertical-first rule, both sides
    while len(sizes) < 40:
        h, w = int(rs.randint(1, 600)), int(rs.randint(1, 900))
        sizes.append((h, w, int(rs.randint(1, 700)), int(rs.randint(1, 700))))
    for h, w, wo, ho in sizes:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((wo, ho), Image.BILINEAR))
        got = dens.prepare_image(torch.from_numpy(img).to(dev), (wo, ho)).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"image {h}x{w} -> {ho}x{wo}")
        m = rs.randint(0, 256, (h, w)).astype(np.uint8)
        near = np.asarray(Image.fromarray(m, mode="L").resize((wo, ho), Image.NEAREST))
        for thr, inv in ((0.5, False), (0.3, True)):
            m01 = ((near.astype(np.float32) / 255.0) > thr)
            m01 = (~m01 if inv else m01).astype(np.uint8)                       # upstream load_mask_resized_np (core/image_utils.py:64-91)
            got_m = dens.prepare_mask(torch.from_numpy(m).to(dev), (wo, ho), threshold=thr, invert=inv).cpu().numpy()
            np.testing.assert_array_equal(got_m, m01, err_msg=f"mask {h}x{w} -> {ho}x{wo} thr {thr} inv {inv}")
        keep = ((near.astype(np.float32) / 255.0) > 0.5).astype(np.uint8)
        black = want * keep[..., None]                                          # apply_mask_to_rgb
        got_b = dens.prepare_image(torch.from_numpy(img).to(dev), (wo, ho), torch.from_numpy(keep).to(dev)).cpu().numpy()
        np.testing.assert_array_equal(got_b, black, err_msg=f"black-out {h}x{w} -> {ho}x{wo}")
    dens.close()
