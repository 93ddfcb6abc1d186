"""Host image I/O for the pipeline driver (upstream core/image_utils.py:12-91).

Same decoding rules as upstream so that the colours sampled by the kernel come from the same
pixels: RGB via PIL, resized to the matcher resolution with ``Image.BILINEAR``; masks converted to
"L", resized with ``Image.NEAREST`` and thresholded at ``> 0.5`` of full scale (1 = keep)."""
from __future__ import annotations

import os
import threading
from collections import OrderedDict
from functools import lru_cache, wraps
from typing import Tuple

import numpy as np

from . import stages

# Budget of the two caches of DECODED (full-resolution) files below.  Upstream only ever caches match-size arrays (~0.8 MB each); a
# decoded 12-24 MP photograph is 36-72 MB, so these caches are bounded by BYTES, not by entries: at the default a camera's decode
# survives until the neighbouring references that list it again have been packed (a few dozen images), never tens of GB.
DECODE_CACHE_BYTES = int(os.environ.get("LFD_DECODE_CACHE_MB", "1024")) << 20


def _byte_bounded_cache(budget_of):
    """``lru_cache`` for functions of hashable arguments returning NumPy arrays, evicting least-recently-used entries while the
    arrays held exceed ``budget_of()`` bytes (an array larger than the whole budget is returned uncached).  Thread-safe: the pack
    threads of the pipeline call the decoders concurrently."""
    def deco(fn):
        store: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
        lock = threading.Lock()
        state = {"bytes": 0, "hits": 0, "misses": 0}

        @wraps(fn)
        def wrapper(*args):
            with lock:
                if args in store:
                    store.move_to_end(args)
                    state["hits"] += 1
                    return store[args]
                state["misses"] += 1
            val = fn(*args)                       # decode outside the lock
            with lock:
                if args not in store and val.nbytes <= budget_of():
                    store[args] = val
                    state["bytes"] += val.nbytes
                    while state["bytes"] > budget_of() and len(store) > 1:
                        _, old = store.popitem(last=False)
                        state["bytes"] -= old.nbytes
            return val

        def cache_clear():
            with lock:
                store.clear()
                state.update(bytes=0, hits=0, misses=0)

        wrapper.cache_clear = cache_clear
        wrapper.cache_info = lambda: dict(state, entries=len(store), budget=budget_of())
        return wrapper
    return deco


def image_dir(scene_root: str, preferred: str) -> str:
    first = os.path.join(scene_root, preferred)
    if os.path.isdir(first):
        return first
    for name in ("images_4", "images_2", "images_8", "images"):
        cand = os.path.join(scene_root, name)
        if os.path.isdir(cand):
            return cand
    raise FileNotFoundError("Could not locate an images directory under scene_root.")


def find_image(root: str, name: str) -> str:
    for cand in (os.path.join(root, name), os.path.join(root, os.path.basename(name))):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError(f"Image '{name}' not found under {root}")


def to_uint8_rgb(rgb01: np.ndarray) -> np.ndarray:
    """[0,1] floats -> u8 with NumPy's round-half-to-even, clipped (upstream core/image_utils.py:24-26)."""
    return np.clip(np.round(np.asarray(rgb01) * 255.0), 0, 255).astype(np.uint8)


@lru_cache(maxsize=4096)
def load_rgb_u8(path: str, size: Tuple[int, int]) -> np.ndarray:
    """(h, w, 3) u8 array of ``path`` resized to ``size=(w, h)``."""
    from PIL import Image
    clock = stages.current()
    with clock.stage("decode", sync=False):
        im = Image.open(path).convert("RGB")
    with clock.stage("prepare", sync=False):
        if im.size != tuple(size):
            im = im.resize(tuple(size), Image.BILINEAR)
        arr = np.asarray(im, dtype=np.uint8)
    arr.setflags(write=False)
    return arr


@lru_cache(maxsize=4096)
def load_mask01(path: str, size: Tuple[int, int], invert: bool = False, threshold: float = 0.5) -> np.ndarray:
    """(h, w) u8 {0,1} mask, 1 = keep."""
    from PIL import Image
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    clock = stages.current()
    with clock.stage("decode", sync=False):
        im = Image.open(path).convert("L")
    with clock.stage("prepare", sync=False):
        if im.size != tuple(size):
            im = im.resize(tuple(size), Image.NEAREST)
        keep = (np.asarray(im, dtype=np.uint8).astype(np.float32) / 255.0) > float(threshold)
        if invert:
            keep = ~keep
        out = keep.astype(np.uint8)
    out.setflags(write=False)
    return out


@_byte_bounded_cache(lambda: DECODE_CACHE_BYTES)
def decode_rgb_u8(path: str) -> np.ndarray:
    """(h, w, 3) u8 array of ``path`` as decoded (no resize): input of the device image preparation (lfd_prepare_image)."""
    from PIL import Image
    with stages.current().stage("decode", sync=False):
        arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    arr.setflags(write=False)
    return arr


@_byte_bounded_cache(lambda: DECODE_CACHE_BYTES)
def decode_mask_l(path: str) -> np.ndarray:
    """(h, w) u8 "L" conversion of a mask file as decoded (no resize, no threshold): input of lfd_prepare_mask."""
    from PIL import Image
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    with stages.current().stage("decode", sync=False):
        arr = np.asarray(Image.open(path).convert("L"), dtype=np.uint8)
    arr.setflags(write=False)
    return arr


def black_out(rgb: np.ndarray, mask01: np.ndarray) -> np.ndarray:
    """Masked pixels become black before matching (upstream core/image_utils.py:69-82)."""
    if mask01.ndim != 2 or mask01.shape != rgb.shape[:2]:
        raise ValueError(f"mask shape {mask01.shape} must match image shape {rgb.shape[:2]}")
    out = np.array(rgb, dtype=np.uint8, copy=True)
    out[mask01 == 0] = 0
    return out
