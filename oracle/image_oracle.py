"""CPU oracle of the image preparation stage (N3)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Upstream prepares every image on the host with Pillow (core/image_utils.py:40-91): ``Image.resize(size, BILINEAR)`` for the
RGB images, ``convert("L")`` + ``Image.resize(size, NEAREST)`` + ``> threshold`` for masks, and blacks out masked pixels
(``apply_mask_to_rgb``, core/pipeline.py:163-171).  The arithmetic lives in Pillow (third-party, not under /root/reference;
pinned: Pillow 12.2.0, libImaging/Resample.c and Geometry.c); this module restates it in NumPy:

  BILINEAR, 8 bits per channel: separable two-pass convolution, horizontal pass first, each pass with per-output-pixel windows
  ``[xmin, xmin+xmax)`` and coefficients computed in f64 (triangle filter of support ``max(scale, 1)``, normalised to sum 1),
  rounded to 22-bit fixed point (``(int)(0.5 + k * 2^22)``); a pass accumulates ``2^21 + sum(pixel * k)`` in int32, shifts
  right by 22 and clamps to [0, 255]; the intermediate image is 8-bit.
  NEAREST: source index ``(int)(offset)`` with ``offset`` ACCUMULATED in f64 (``+= scale`` per output pixel) from ``scale/2``.

``tests/test_image_prep.py`` requires this restatement to equal Pillow bit for bit on this machine (and the committed golden
vectors captured through upstream's own functions); the HIP kernels are then compared with it on the GPU box.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the triangle filter over the whole input
    (box = (0, in_size)).  Returns ``bounds (out,2) int32 [xmin, count]``, ``kk (out, ksize) int32`` and ``ksize``."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (double)(in1 - in0) / outSize, box coordinates are floats
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            wgt = 1.0 - v if v < 1.0 else 0.0
            k[x] = wgt
            ww += wgt
        if ww != 0.0:
            for x in range(xmax):
                k[x] /= ww
        for x in range(ksize):
            p = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if k[x] < 0 else int(0.5 + p)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One 8-bit resampling pass along ``axis`` (1 = horizontal, 0 = vertical) of an (h, w, c) u8 image."""
    src = img.astype(np.int64)
    out_n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_n
    out = np.empty(shape, np.uint8)
    for o in range(out_n):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(np.delete(np.array(img.shape), axis), 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(cnt):
            acc = acc + (src[:, lo + t, :] if axis == 1 else src[lo + t, :, :]) * int(kk[o, t])
        val = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o, :] = val
        else:
            out[o, :, :] = val
    return out


def resize_bilinear_u8(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """``Image.fromarray(img).resize(size_wh, Image.BILINEAR)`` for an (h, w, 3) u8 image (core/image_utils.py:85-91)."""
    w_out, h_out = int(size_wh[0]), int(size_wh[1])
    h_in, w_in = img.shape[:2]
    cur = img
    if h_in > w_in * 100 and h_out < h_in:
        # Pillow's Image.resize (PIL/Image.py, "if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]"): an image more than 100 times
        # taller than wide that shrinks vertically is resampled VERTICALLY FIRST, then horizontally - two resize calls, the intermediate
        # image rounded to 8 bits like every pass
        bv, kv, _ = bilinear_coeffs(h_in, h_out)
        cur = _pass(cur, bv, kv, axis=0)
        if w_out != w_in:
            bh, kh, _ = bilinear_coeffs(w_in, w_out)
            cur = _pass(cur, bh, kh, axis=1)
        return np.ascontiguousarray(cur)
    if w_out != w_in:
        bh, kh, _ = bilinear_coeffs(w_in, w_out)
        if h_out != h_in:           # Pillow resamples only the rows the vertical pass will read
            bv, _, _ = bilinear_coeffs(h_in, h_out)
            first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
            part = _pass(cur[first:last], bh, kh, axis=1)
            cur = np.zeros((h_in, w_out, img.shape[2]), np.uint8)
            cur[first:last] = part
        else:
            cur = _pass(cur, bh, kh, axis=1)
    if h_out != h_in:
        bv, kv, _ = bilinear_coeffs(h_in, h_out)
        cur = _pass(cur, bv, kv, axis=0)
    return np.ascontiguousarray(cur)


def nearest_indices(in_size: int, out_size: int) -> np.ndarray:
    """Source index of every output pixel for ``Image.resize(size, NEAREST)`` (Pillow's affine scale path, Geometry.c
    ``ImagingScaleAffine``): the offset starts at ``scale * 0.5`` and is ACCUMULATED in f64; index = (int)offset."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    xo = 0.0 + scale * 0.5
    idx = np.zeros(out_size, np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        idx[x] = min(max(xin, 0), in_size - 1)
        xo += scale
    return idx


def mask01_resized(mask_l: np.ndarray, size_wh: Tuple[int, int], threshold: float = 0.5, invert: bool = False) -> np.ndarray:
    """``load_mask_resized_np`` after the "L" conversion (core/image_utils.py:40-66): NEAREST resize, ``> threshold`` of full scale."""
    w_out, h_out = int(size_wh[0]), int(size_wh[1])
    h_in, w_in = mask_l.shape
    arr = mask_l
    if (w_in, h_in) != (w_out, h_out):
        arr = mask_l[nearest_indices(h_in, h_out)][:, nearest_indices(w_in, w_out)]
    keep = (arr.astype(np.float32) / 255.0) > float(threshold)
    if invert:
        keep = ~keep
    return keep.astype(np.uint8)


def mask_threshold_lut(threshold: float = 0.5, invert: bool = False) -> np.ndarray:
    """The 256-entry table of ``(v / 255.0 as f32) > threshold`` the device applies per byte."""
    keep = (np.arange(256, dtype=np.uint8).astype(np.float32) / 255.0) > float(threshold)
    if invert:
        keep = ~keep
    return keep.astype(np.uint8)


def black_out(rgb: np.ndarray, mask01: np.ndarray) -> np.ndarray:
    out = np.array(rgb, dtype=np.uint8, copy=True)
    out[mask01 == 0] = 0
    return out
